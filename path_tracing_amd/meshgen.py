"""Deterministic stand-in meshes and binary STL I/O.

The reference renders a subdivided Suzanne that is not distributed with it (its
``inputs/`` directory is git-ignored), so benchmarks and tests use a procedurally
generated object of similar size (about 2.7 x 1.7 x 2 units around the origin) standing on
a floor.  Vertices shared between triangles are bit-identical, which the vertex-normal
pass relies on (reference stlreader.h:13), and files are binary STL with a zero attribute
count (reference stlreader.cpp:35-38).
"""
from __future__ import annotations

import numpy as np

_STL_DTYPE = np.dtype([("normal", "<f4", (3,)), ("verts", "<f4", (9,)), ("attr", "<u2")])


def _icosahedron():
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array(
        [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t],
         [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array(
        [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
         [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
         [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    return v, f


def _subdivide(v, f):
    """One 1:4 subdivision with shared edge midpoints pushed back to the unit sphere."""
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
    e.sort(axis=1)
    key = e[:, 0] * np.int64(v.shape[0]) + e[:, 1]
    uniq, inverse = np.unique(key, return_inverse=True)
    a = uniq // v.shape[0]
    b = uniq % v.shape[0]
    mid = v[a] + v[b]
    mid /= np.linalg.norm(mid, axis=1, keepdims=True)
    base = v.shape[0]
    v2 = np.concatenate([v, mid], axis=0)
    n = f.shape[0]
    m01 = base + inverse[0:n]
    m12 = base + inverse[n:2 * n]
    m20 = base + inverse[2 * n:3 * n]
    f2 = np.concatenate([
        np.stack([f[:, 0], m01, m20], axis=1),
        np.stack([f[:, 1], m12, m01], axis=1),
        np.stack([f[:, 2], m20, m12], axis=1),
        np.stack([m01, m12, m20], axis=1)], axis=0)
    return v2, f2


def bumpy_object(level: int):
    """Displaced, anisotropically scaled icosphere: 20 * 4**level triangles.

    Returns (vertices float32 [V,3], faces int64 [F,3]) with outward winding.
    """
    v, f = _icosahedron()
    for _ in range(level):
        v, f = _subdivide(v, f)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    r = (1.0 + 0.10 * np.sin(5.0 * x + 0.3) * np.sin(4.0 * y) * np.cos(3.0 * z + 0.7)
         + 0.05 * np.sin(11.0 * x) * np.cos(13.0 * y + 1.1) + 0.03 * np.cos(23.0 * z) * np.sin(19.0 * x * y))
    p = v * r[:, None] * np.array([1.35, 0.85, 1.0])
    return p.astype(np.float32), f


def floor_grid(n: int, half_extent: float = 4.0, z: float = -1.15):
    """n x n quads (2 n^2 triangles) in the plane z, facing +z."""
    lin = np.linspace(-half_extent, half_extent, n + 1)
    gx, gy = np.meshgrid(lin, lin, indexing="xy")
    v = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, z)], axis=1).astype(np.float32)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="xy")
    a = (j * (n + 1) + i).ravel()
    b = a + 1
    c = a + (n + 1)
    d = c + 1
    f = np.concatenate([np.stack([a, b, d], axis=1), np.stack([a, d, c], axis=1)], axis=0)
    return v, f.astype(np.int64)


def standin_scene(level: int, floor_n: int | None = None) -> np.ndarray:
    """Triangle soup float32 [N, 9] (p0 p1 p2) of the object plus its floor.

    level 3 -> 1 280 object triangles, level 8 -> 1 310 720, level 10 -> 20 971 520.
    """
    if floor_n is None:
        floor_n = max(2, 2 ** max(level - 1, 1))
    ov, of = bumpy_object(level)
    fv, ff = floor_grid(floor_n)
    tris = np.concatenate([ov[of].reshape(-1, 9), fv[ff].reshape(-1, 9)], axis=0)
    return np.ascontiguousarray(tris, dtype=np.float32)


def write_stl(path: str, tris: np.ndarray) -> None:
    tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
    rec = np.zeros(tris.shape[0], dtype=_STL_DTYPE)
    rec["verts"] = tris
    e1 = tris[:, 3:6] - tris[:, 0:3]
    e2 = tris[:, 6:9] - tris[:, 0:3]
    n = np.cross(e1, e2)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    rec["normal"] = np.where(ln > 0, n / np.maximum(ln, 1e-30), 0).astype(np.float32)
    with open(path, "wb") as fp:
        fp.write(b"path_tracing_amd stand-in mesh".ljust(80, b" "))
        fp.write(np.uint32(tris.shape[0]).tobytes())
        fp.write(rec.tobytes())


def read_stl(path: str) -> np.ndarray:
    with open(path, "rb") as fp:
        fp.read(80)
        n = int(np.frombuffer(fp.read(4), dtype="<u4")[0])
        rec = np.frombuffer(fp.read(n * _STL_DTYPE.itemsize), dtype=_STL_DTYPE, count=n)
    return np.ascontiguousarray(rec["verts"], dtype=np.float32)
