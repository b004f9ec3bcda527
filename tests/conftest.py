"""Shared fixtures.  Tests marked `gpu` need an MI355X and call through the C ABI of
libpt_hip.so; everything else runs on the CPU (oracle, host-side builder, lane emulation)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import orc  # noqa: E402  (the CPU oracle: checker only)
import path_tracing_amd as pta  # noqa: E402


def rays_sphere(n, seed=1234, radius=3.0):
    """Origins uniform on a sphere, aimed at points uniform in the unit ball (incoherent set)."""
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = radius * o / np.linalg.norm(o, axis=1, keepdims=True)
    t = rng.normal(size=(n, 3))
    t = t / np.linalg.norm(t, axis=1, keepdims=True) * rng.random((n, 1)) ** (1.0 / 3.0)
    d = (t - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o.astype(np.float32), d.astype(np.float32)


def rays_inside(n, seed=5):
    rng = np.random.default_rng(seed)
    o = (rng.random((n, 3)) * 2 - 1).astype(np.float32) * np.array([1.3, 0.8, 1.0], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def rays_axis_parallel(n, seed=6):
    rng = np.random.default_rng(seed)
    o = (rng.random((n, 3)) * 4 - 2).astype(np.float32)
    d = np.zeros((n, 3), np.float32)
    d[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-1.0, 1.0], n)
    return o, d


def rays_at_vertices(tris, n, seed=7, radius=3.0):
    """Rays aimed exactly at mesh vertices and edge midpoints (shared-edge / crack cases)."""
    rng = np.random.default_rng(seed)
    t = tris.reshape(-1, 3, 3)
    k = rng.integers(0, t.shape[0], n)
    a = t[k, rng.integers(0, 3, n)]
    b = t[k, rng.integers(0, 3, n)]
    target = np.where(rng.random((n, 1)) < 0.5, a, 0.5 * (a + b))
    o = rng.normal(size=(n, 3))
    o = (radius * o / np.linalg.norm(o, axis=1, keepdims=True)).astype(np.float32)
    d = (target - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def rays_hostile(tris, n, seed=8):
    """Rays that stress the arithmetic contract rather than the geometry: one zero direction
    component (plane-parallel), negative zeros, denormal and huge components, origins far away
    (beyond the range where the cheaper slab test may be used), origins exactly on vertices and
    on the floor plane with grazing directions, and non-finite inputs (NaN / infinite origins),
    for which the reference's comparisons all come out false and the walk still terminates."""
    rng = np.random.default_rng(seed)
    o, d = rays_sphere(n, seed=seed + 100)
    o = o.copy(); d = d.copy()
    k = np.arange(n)
    cls = k % 8
    ax = rng.integers(0, 3, n)
    m = cls == 0                                   # one component exactly zero
    d[m, ax[m]] = 0.0
    m = cls == 1                                   # ... or negative zero
    d[m, ax[m]] = -0.0
    m = cls == 2                                   # denormal component
    d[m, ax[m]] = np.float32(1e-41) * rng.choice([-1.0, 1.0], m.sum()).astype(np.float32)
    m = cls == 3                                   # far origins, still aimed at the scene
    scale = rng.choice(np.array([1e3, 1e8, 1e20, 1e31, 3e37], np.float32), m.sum())
    o[m] = o[m] * scale[:, None]
    m = cls == 4                                   # origin exactly on a mesh vertex
    t = tris.reshape(-1, 3, 3)
    o[m] = t[rng.integers(0, t.shape[0], m.sum()), rng.integers(0, 3, m.sum())]
    m = cls == 5                                   # in the plane of the lowest vertices, grazing
    o[m, 2] = tris.reshape(-1, 3)[:, 2].min()
    d[m, 2] = 0.0
    m = cls == 6                                   # non-finite origins
    bad = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), m.sum())
    o[m, ax[m]] = bad
    m = cls == 7                                   # huge direction components (not unit length)
    d[m] = d[m] * rng.choice(np.array([1e-30, 1e-10, 1e10, 1e30], np.float32), m.sum())[:, None]
    return o.astype(np.float32), d.astype(np.float32)


def oversized_leaf_mesh(k=24, per_leaf=40):
    """k*k groups of `per_leaf` triangles that all have the SAME bounding box (they share the box
    diagonal, the third vertex slides across it), so no split separates a group: the builder
    makes each group one leaf (reference kdtree.cpp:206-208) of more than 15 triangles, the size
    at which a child reference no longer holds the leaf's triangle count."""
    s = np.linspace(0.05, 0.95, per_leaf, dtype=np.float32)
    t = np.zeros((per_leaf, 9), np.float32)
    t[:, 3:6] = 1.0
    t[:, 6] = s
    t[:, 7] = 1 - s
    t[:, 8] = 0.5
    shifts = np.stack(np.meshgrid(np.arange(k), np.arange(k), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float32) * 1.5
    return np.concatenate([t + np.tile(np.array([sx, sy, 0], np.float32), 3) for sx, sy in shifts], axis=0)


def rays_over_grid(n, extent, seed=4):
    """Rays from above and below a slab [0, extent]^2 x [0, 1], tilted a little."""
    rng = np.random.default_rng(seed)
    o = np.empty((n, 3), np.float32)
    o[:, :2] = rng.random((n, 2)).astype(np.float32) * extent
    o[:, 2] = np.where(rng.random(n) < 0.5, 5.0, -5.0)
    d = rng.normal(size=(n, 3)).astype(np.float32) * np.float32(0.15)
    d[:, 2] = -np.sign(o[:, 2])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def all_ray_sets(tris, n=20000):
    cam = orc.cli_camera()
    sets = {
        "primary": orc.camera_rays(cam, 160, 120),
        "sphere": rays_sphere(n),
        "inside": rays_inside(n),
        "axis": rays_axis_parallel(n),
        "vertices": rays_at_vertices(tris, n),
        "hostile": rays_hostile(tris, n),
    }
    return sets


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_hits_equal(a, b, what=""):
    names = ("index", "t", "u", "v")
    for x, y, nm in zip(a, b, names):
        bad = np.nonzero(bits(x) != bits(y))[0]
        assert bad.size == 0, f"{what}: {bad.size} rays differ in {nm}, first {bad[:5]}: {x[bad[:5]]} vs {y[bad[:5]]}"


class SceneSet:
    def __init__(self, level, floor_n=None):
        self.tris = pta.meshgen.standin_scene(level, floor_n)
        self.oracle = orc.Scene(self.tris, threads=8)
        self.host = pta.HostScene(self.tris, threads=8)
        self._gpu = None

    @property
    def gpu(self):
        if self._gpu is None:
            self._gpu = pta.GpuScene(self.host, 0)
        return self._gpu


@pytest.fixture(scope="session")
def scene_small():
    return SceneSet(3)          # 1 312 triangles


@pytest.fixture(scope="session")
def scene_medium():
    return SceneSet(5)          # 20 992 triangles


@pytest.fixture(scope="session")
def scene_tiny():
    return SceneSet(0)          # 28 triangles: tree of depth 3


def load_emul():
    path = os.path.join(ROOT, "tests", "native", "_build", "libtraverse_emul.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", ROOT, "-s", "native-tests"])
    E = C.CDLL(path)
    vp = C.c_void_p
    E.emul_raycast_closest.argtypes = [vp, C.c_uint64] + [vp] * 7
    E.emul_raycast_shadow.argtypes = [vp, C.c_uint64] + [vp] * 4
    E.emul_raycast_shadow_q.argtypes = [vp, C.c_uint64] + [vp] * 5
    E.emul_raycast_occluded_q.argtypes = [vp, C.c_uint64] + [vp] * 4
    E.emul_work_counts.argtypes = [vp, C.c_uint64, vp, vp]
    return E
