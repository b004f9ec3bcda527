"""Meshes that stress the arithmetic contract rather than the rays: coordinates beyond the range
where the cheaper slab test may be used (every lane then takes the NaN-exact one) and whose
products overflow, a mesh so small that the triangle test's absolute threshold cuts hits, zero-area and coincident triangles
(exact ties), and one mesh spanning 30 orders of magnitude whose root box is beyond that range while
its small triangles are still hit; and random triangle soups of 1 to 5000 triangles.  Host builder = oracle tree, lane
emulation = oracle results on the CPU; the HIP path = oracle results through the C ABI (gpu)."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_hits_equal, load_emul, orc, pta, rays_hostile, rays_sphere


def hostile_meshes():
    base = pta.meshgen.standin_scene(3).astype(np.float32)
    rng = np.random.default_rng(21)
    meshes = {}
    meshes["huge"] = (base * np.float32(1e31), 1e31)            # root box not "tame"; edge products overflow
    meshes["tiny"] = (base * np.float32(2e-4), 2e-4)            # determinants on both sides of the 1e-8 threshold (utils.cpp:142): a third of the hits are cut
    t = base.reshape(-1, 3, 3)
    k = rng.integers(0, len(t), 60)
    two_equal = t[k[:20]].copy(); two_equal[:, 1] = two_equal[:, 0]          # zero-area: two vertices equal
    point = t[k[20:40]].copy(); point[:, 1] = point[:, 0]; point[:, 2] = point[:, 0]   # ... all three equal
    dup = t[k[40:]].copy()                                                    # coincident copies: exact ties
    meshes["degenerate"] = (np.concatenate([base, two_equal.reshape(-1, 9), point.reshape(-1, 9), dup.reshape(-1, 9),
                                            dup.reshape(-1, 9)]).astype(np.float32), 1.0)
    far = np.array([[2e30, 0, 0, 0, 2e30, 0, 0, 0, 2e30], [-1e25, -1e25, -5, 1e25, -1e25, -5, 0, 1e25, -5]], np.float32)   # 2e30: the root box is not "tame"
    meshes["mixed_scale"] = (np.concatenate([base, far]).astype(np.float32), 1.0)
    # Random triangle soups: boxes overlap everywhere, so the walk takes the "origin inside both
    # children" branch (kdtree.cpp:324-342) and the far-side rule far more often than on a surface.
    for n in (1, 2, 9, 300, 5000):
        c = rng.random((n, 1, 3)).astype(np.float32) * 2 - 1
        size = np.float32(10.0) ** rng.uniform(-2.5, 0.0, (n, 1, 1)).astype(np.float32)
        soup = c + (rng.normal(size=(n, 3, 3)).astype(np.float32) * size)
        if n >= 300:   # long thin slivers across the whole cloud
            soup[:n // 10, 1] = soup[:n // 10, 0] + np.float32(1e-4) * rng.normal(size=(n // 10, 3)).astype(np.float32)
            soup[:n // 10, 2] = -soup[:n // 10, 0]
        meshes["soup%d" % n] = (soup.reshape(-1, 9).astype(np.float32), 1.0)
    return meshes


MESH_NAMES = ["huge", "tiny", "degenerate", "mixed_scale", "soup1", "soup2", "soup9", "soup300", "soup5000"]


def ray_sets(tris, scale, n=6000):
    o, d = rays_sphere(n, seed=31)
    ho, hd = rays_hostile(tris, n, seed=32)
    s = np.float32(scale)
    with np.errstate(over="ignore", invalid="ignore"):
        return {"sphere": (o * s, d), "hostile": (ho * s, hd)}


def shadow_dist(n):
    """Light distances on both sides of 1 (below 1 the reference's answer depends on the closest hit's v)."""
    rng = np.random.default_rng(33)
    return np.where(rng.random(n) < 0.5, rng.random(n), 1.0 + 4.0 * rng.random(n)).astype(np.float32)


@pytest.mark.parametrize("name", MESH_NAMES)
def test_host_tree_and_lane_walk(name):
    tris, scale = hostile_meshes()[name]
    host = pta.HostScene(tris)
    oracle = orc.Scene(tris)
    (n1, l1), (n2, l2) = oracle.tree_dump(), host.tree_dump()
    assert np.array_equal(n1.view(np.uint8), n2.view(np.uint8)) and np.array_equal(l1, l2), "the host builder's tree differs from the oracle's"
    E = load_emul()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for rname, (o, d) in ray_sets(tris, scale).items():
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        n = len(o)
        idx = np.empty(n, np.int32); t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32)
        worst = E.emul_raycast_closest(C.byref(host.flat), n, p(o), p(d), p(idx), p(t), p(u), p(v), None)
        assert 0 <= worst <= 60
        assert_hits_equal((idx, t, u, v), oracle.raycast(o, d), f"{name}/{rname}")
        dist = shadow_dist(n)
        vis = np.empty(n, np.uint8)
        assert E.emul_raycast_shadow(C.byref(host.flat), n, p(o), p(d), p(dist), p(vis)) >= 0
        assert np.array_equal(vis, oracle.shadow_visible(o, d, dist)), f"{name}/{rname} shadow"
        # the any-hit walks on quantised interior boxes (where the mesh gets them): same answers
        rc = E.emul_raycast_shadow_q(C.byref(host.flat), n, p(o), p(d), p(dist), p(vis), None)
        assert rc in (0, -2)
        if rc == 0:
            assert np.array_equal(vis, oracle.shadow_visible(o, d, dist)), f"{name}/{rname} shadow on q-nodes"
            want = oracle.raycast(o, d)
            bound = np.where(want[0] >= 0, want[1] * np.float32(1.5), np.float32(1.0)).astype(np.float32)
            assert E.emul_raycast_occluded_q(C.byref(host.flat), n, p(o), p(d), p(bound), p(vis)) == 0
            assert np.array_equal(vis, oracle.shadow_visible(o, d, bound, bounded=True)), f"{name}/{rname} bounded on q-nodes"


@pytest.mark.gpu
@pytest.mark.parametrize("name", MESH_NAMES)
@pytest.mark.parametrize("option", ["", "PT_QNODES", "PT_DENSE_SHADOW", "PT_LEAFWISE", "PT_LEAFWISE+PT_QNODES"])
def test_gpu_matches_oracle(name, option, monkeypatch):
    """option: the any-hit walks on quantised interior nodes, with deferred dense triangle tests, with
    leaves tested by the whole wave, or the last with its any-hit walks on quantised interior nodes."""
    tris, scale = hostile_meshes()[name]
    host = pta.HostScene(tris)
    oracle = orc.Scene(tris)
    for name_value in filter(None, option.split("+")):
        monkeypatch.setenv(*(name_value.split("=") if "=" in name_value else (name_value, "1")))
    gpu = pta.GpuScene(host, 0)
    some_hit = False
    for rname, (o, d) in ray_sets(tris, scale).items():
        want = oracle.raycast(o, d)
        assert_hits_equal(gpu.raycast_closest(o, d), want, f"{name}/{rname}")
        some_hit = some_hit or (want[0] >= 0).any()
        dist = shadow_dist(len(o))
        assert np.array_equal(gpu.raycast_shadow(o, d, dist), oracle.shadow_visible(o, d, dist)), f"{name}/{rname} shadow"
        bound = np.where(want[0] >= 0, want[1] * np.float32(1.5), np.float32(1.0)).astype(np.float32)
        assert np.array_equal(gpu.raycast_shadow(o, d, bound, bounded=True), oracle.shadow_visible(o, d, bound, bounded=True)), f"{name}/{rname} bounded"
    # every product of two edges of the huge mesh overflows, so there every test is a miss on both sides
    assert some_hit or name == "huge", "the ray sets never hit this mesh: the case tests nothing"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["soup300", "degenerate"])
def test_gpu_render_matches_oracle(name):
    """The whole bounce loop on a mesh where mirror paths run deep (a soup reflects into itself) and
    on one with exact ties: same ray counts, same pass counts, radiance within the summation-order
    tolerance of the render parity test."""
    tris, _ = hostile_meshes()[name]
    oracle = orc.Scene(tris)
    gpu = pta.GpuScene(pta.HostScene(tris), 0)
    W, H, S = 64, 48, 3
    want_rgb, want_p, want_counts = oracle.render(orc.cli_camera(dof_aperture=0.1, dof_distance=4.5), orc.cli_lights(), W, H, None, 0, S, seed=9)
    film = pta.Film(gpu, W, H)
    st = film.render(pta.cli_camera(dof_aperture=0.1, dof_distance=4.5), pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=9)
    rgb, passes = film.read()
    assert np.array_equal(passes, want_p)
    assert st.closest_rays == int(want_counts[0]) and st.shadow_rays == int(want_counts[1])
    assert st.closest_rays > 1.5 * W * H * S, "paths should bounce on this mesh"
    err = np.abs(rgb - want_rgb)
    assert (err <= 1e-5 * np.maximum(np.abs(want_rgb), 1.0)).all(), f"max err {err.max()}"
