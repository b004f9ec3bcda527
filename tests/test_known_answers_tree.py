"""Known answers for the tree BUILD (reference kdtree.cpp:32-262) and for the VISIT ORDER of the walk
(kdtree.cpp:307-394), worked out by hand from the reference's source on small meshes whose
coordinates binary32 represents exactly.  They pin the oracle, the host-side builder, the lane logic
and (in the GPU half) the HIP path to the reference's rules independently of any of them:

  build   candidates are the box maxima of the node's triangles taken in by-minimum-x order (:107-108);
          count_low = #(minimum <= s), count_high = #(maximum > s) (:110-127); score = the larger of
          the two, the FIRST candidate with the smallest score wins, axis 0 before 1 before 2
          (:129-143); a triangle goes low iff its minimum <= s (:172-183); a leaf lists its triangles in
          by-minimum-x order (:34, :68); at most 8 triangles make a leaf (:67); a split that leaves one
          side with everything makes a leaf instead (:206-208).
  walk    both children are entered low first, high second when the origin lies in both children's
          extents along the split axis (:310-342); otherwise the near child first, the far child only
          if the near hit lies beyond the far child's slab (:315-316, :352-394); a later hit replaces
          an earlier one only if it is STRICTLY nearer.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import load_emul, orc, pta


def tri2d(ax, ay, bx, by, cx, cy):
    return [ax, ay, 0.0, bx, by, 0.0, cx, cy, 0.0]


# ---------------------------------------------------------------------------------------------------
# TEN: ten triangles in the plane z = 0.
#   P = (0,0) (4,0) (0,4)          box x [0, 4],       y [0, 4]
#   Q = (1,-1) (5,-1) (1,3)        box x [1, 5],       y [-1, 3]
#   F1..F4: x [-0.5, 0.5], [-0.375, 0.5], [-0.25, 0.5], [-0.125, 0.5];  y [100,101], [102,103], [104,105], [106,107]
#   G1..G4: x [4.5, 5], [4.5625, 5], [4.625, 5], [4.6875, 5];           y as F1..F4
# By-minimum-x order: F1 F2 F3 F4 P Q G1 G2 G3 G4.  Ten triangles > 8, every one is a candidate (< 100).
# Axis 0, candidates in that order: s = 0.5 (F1): minima <= 0.5: F1..F4, P = 5; maxima > 0.5: P, Q, G1..G4
# = 6; score 6.  F2..F4 the same.  s = 4 (P): 6 and 5, score 6 -- not smaller.  s = 5: 10 and 0.  Axis 1
# (maxima in y: 101, 103, 105, 107, 4, 3, 101, ...): s = 101: minima <= 101: Q, P, F1, G1 = 4, maxima >
# 101: 6, score 6; s = 103: 6 and 4; s = 105: 8; s = 107: 10; s = 4: 2 and 8; s = 3: 2 and 9 -- none
# below 6.  Axis 2: every box is [0, 0], score 10.  So the root splits axis 0 at 0.5; low = the
# triangles whose minimum in x is <= 0.5 = F1 F2 F3 F4 P, high = Q G1 G2 G3 G4, five each: leaves.
P_TRI = tri2d(0, 0, 4, 0, 0, 4)
Q_TRI = tri2d(1, -1, 5, -1, 1, 3)
F = [tri2d(a, b, 0.5, b, a, b + 1) for a, b in ((-0.5, 100), (-0.375, 102), (-0.25, 104), (-0.125, 106))]
G = [tri2d(a, b, 5, b, a, b + 1) for a, b in ((4.5, 100), (4.5625, 102), (4.625, 104), (4.6875, 106))]
# mesh order, deliberately not the tree's:  G3 P F2 Q G1 F4 F1 G4 F3 G2
TEN = np.array([G[2], P_TRI, F[1], Q_TRI, G[0], F[3], F[0], G[3], F[2], G[1]], np.float32)
IDX_P, IDX_Q = 1, 3
TEN_LEAVES = [6, 2, 8, 5, 1,   3, 4, 9, 0, 7]          # F1 F2 F3 F4 P | Q G1 G2 G3 G4
# pre-order dump rows: [leaf, axis, split, box min xyz, box max xyz, triangles]
TEN_ROOT = [0, 0, 0.5, -0.5, -1, 0, 5, 107, 0, 0]
TEN_LOW_BOX = [-0.5, 0, 0, 4, 107, 0]
TEN_HIGH_BOX = [1, -1, 0, 5, 107, 0]

# Rays through X = (1.5, 0.5, 0), which both P and Q cover; directions are NOT normalised (t is in
# units of the given direction), so every product below is exact.  For P (e1 = (4,0,0), e2 = (0,4,0),
# det = 32 with each of these directions) u = 0.375, v = 0.125; for Q u = 0.125, v = 0.375; t = 1 for both.
#   1. origin x = 1.5, straight down: 1 < 1.5 and 1.5 <= 4, the origin lies in both extents: low (P)
#      first, Q's equal t does not replace it.
#   2. origin x = 3.5, travelling towards -x: still inside both extents, so low first although the
#      ray comes from the high side -> P.
#   3. origin x = 4.5 > 4 = low's maximum: near = high; Q is hit at t = 1; t_far = (4 - 4.5) / -3 =
#      0.1667 < 1, so low is entered too, and P's equal t does not replace Q.
#   4. origin x = -1.5: 1 < -1.5 is false, near = low; P at t = 1; t_far = (1 + 1.5) / 3 = 0.8333 < 1,
#      high is entered, Q's equal t does not replace P.
ORDER_RAYS = [
    ((1.5, 0.5, 2.0), (0.0, 0.0, -2.0), (IDX_P, 1.0, 0.375, 0.125)),
    ((3.5, 0.5, 2.0), (-2.0, 0.0, -2.0), (IDX_P, 1.0, 0.375, 0.125)),
    ((4.5, 0.5, 2.0), (-3.0, 0.0, -2.0), (IDX_Q, 1.0, 0.125, 0.375)),
    ((-1.5, 0.5, 2.0), (3.0, 0.0, -2.0), (IDX_P, 1.0, 0.375, 0.125)),
    # and through (0.5, 0.5, 0), which only P covers (Q starts at x = 1): u = 0.125, v = 0.125
    ((0.5, 0.5, 2.0), (0.0, 0.0, -2.0), (IDX_P, 1.0, 0.125, 0.125)),
    # through (4.5, -0.5, 0), which only Q covers: T = (3.5, 0.5, 2): u = 3.5*8/32 = 0.875, v = 0.125
    ((4.5, -0.5, 2.0), (0.0, 0.0, -2.0), (IDX_Q, 1.0, 0.875, 0.125)),
]

# ---------------------------------------------------------------------------------------------------
# DIAG: twenty triangles along the diagonal, triangle i = (2i,2i) (2i+1,2i) (2i,2i+1): box [2i, 2i+1] in x
# and in y.  Root (20 > 8): candidate k has s = 2k+1, count_low = k+1, count_high = 19-k, score =
# max(k+1, 19-k), smallest (10) at k = 9 first: split 19.  Axis 1 gives the very same scores and loses
# to axis 0 (a later candidate must be strictly better).  Both halves hold ten (> 8) and split again at
# their fifth triangle: 9 and 29.  Four leaves of five.
DIAG = np.array([tri2d(2 * i, 2 * i, 2 * i + 1, 2 * i, 2 * i, 2 * i + 1) for i in range(20)], np.float32)
DIAG_PERM = [(7 * i + 3) % 20 for i in range(20)]          # mesh position j holds triangle DIAG_PERM[j]


def diag_dump():
    def box(a, b):
        return [2 * a, 2 * a, 0, 2 * b + 1, 2 * b + 1, 0]
    rows = [[0, 0, 19] + box(0, 19) + [0],
            [0, 0, 9] + box(0, 9) + [0], [1, None, None] + box(0, 4) + [5], [1, None, None] + box(5, 9) + [5],
            [0, 0, 29] + box(10, 19) + [0], [1, None, None] + box(10, 14) + [5], [1, None, None] + box(15, 19) + [5]]
    where = {t: j for j, t in enumerate(DIAG_PERM)}
    return rows, [where[t] for t in range(20)]

# ---------------------------------------------------------------------------------------------------
# NINE: nine copies of one triangle.  Every candidate puts all nine on the low side (their minimum
# <= their maximum), so the split does not improve and the root becomes a leaf of nine although nine
# > 8 (:206-208); equal keys keep their order in libstdc++'s sort of nine elements (plain insertion).
NINE = np.array([tri2d(0, 0, 1, 0, 0, 1)] * 9, np.float32)


def check_dump(nodes, leaves, want_rows, want_leaves, who):
    assert len(nodes) == len(want_rows), who
    for got, want in zip(nodes.tolist(), want_rows):
        for k, w in enumerate(want):
            if w is not None:
                assert got[k] == w, (who, got, want)
    assert leaves.tolist() == want_leaves, who


def builders():
    return [("oracle", lambda t: orc.Scene(t)), ("host builder", lambda t: pta.HostScene(t)),
            ("host builder, 5 threads", lambda t: pta.HostScene(t, threads=5))]


def test_build_known_answers():
    for who, make in builders():
        nodes, leaves = make(TEN).tree_dump()
        check_dump(nodes, leaves, [TEN_ROOT, [1, None, None] + TEN_LOW_BOX + [5], [1, None, None] + TEN_HIGH_BOX + [5]], TEN_LEAVES, who)
        rows, order = diag_dump()
        nodes, leaves = make(np.ascontiguousarray(DIAG[DIAG_PERM])).tree_dump()
        check_dump(nodes, leaves, rows, order, who)
        nodes, leaves = make(NINE).tree_dump()
        check_dump(nodes, leaves, [[1, None, None, 0, 0, 0, 1, 1, 0, 9]], list(range(9)), who)


def order_arrays():
    o = np.array([c[0] for c in ORDER_RAYS], np.float32)
    d = np.array([c[1] for c in ORDER_RAYS], np.float32)
    want = [np.array([c[2][k] for c in ORDER_RAYS], dt) for k, dt in enumerate((np.int32, np.float32, np.float32, np.float32))]
    return o, d, want


def check_hits(got, want, who):
    for g, w, name in zip(got, want, ("index", "t", "u", "v")):
        assert np.array_equal(np.asarray(g), w), (who, name, np.asarray(g), w)


def test_visit_order_known_answers_oracle_and_lane_logic():
    o, d, want = order_arrays()
    check_hits(orc.Scene(TEN).raycast(o, d), want, "oracle")
    host = pta.HostScene(TEN)
    E = load_emul()
    n = len(o)
    idx, t, u, v, pos = np.empty(n, np.int32), np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert E.emul_raycast_closest(C.byref(host.flat), n, p(o), p(d), p(idx), p(t), p(u), p(v), p(pos)) >= 0
    check_hits((idx, t, u, v), want, "lane logic on the flat arrays")
    # the exhaustive search has no visit order: on the four tied rays it keeps the first triangle of
    # the MESH that is hit (P, index 1, comes before Q, index 3) -- the walk's answer for ray 3 is not
    # "the nearest triangle, lowest index", it is the reference's visit order
    brute = orc.Scene(TEN).raycast_brute(o, d)
    assert brute[0].tolist() == [IDX_P, IDX_P, IDX_P, IDX_P, IDX_P, IDX_Q]
    assert want[0][2] == IDX_Q


@pytest.mark.gpu
def test_visit_order_known_answers_gpu(monkeypatch):
    o, d, want = order_arrays()
    host = pta.HostScene(TEN)
    check_hits(pta.GpuScene(host, 0).raycast_closest(o, d), want, "GPU closest hit")
    # the other trace kernels (several triangles per round; leaves tested by the whole wave) follow the same order
    for name, value in (("PT_TRIS_PER_ROUND", "2"), ("PT_TRIS_PER_ROUND", "3"), ("PT_LEAFWISE", "1")):
        monkeypatch.setenv(name, value)
        gpu = pta.GpuScene(host, 0)
        monkeypatch.delenv(name)
        check_hits(gpu.raycast_closest(o, d), want, f"GPU closest hit, {name}={value}")
    # a batch large enough for the persistent kernel's refill path: the same six rays many times over
    reps = 4096
    got = pta.GpuScene(host, 0).raycast_closest(np.tile(o, (reps, 1)), np.tile(d, (reps, 1)))
    check_hits(got, [np.tile(w, reps) for w in want], "GPU closest hit, 24 576 rays")


# ---------------------------------------------------------------------------------------------------
# FOLD: per-triangle data (Triangle::Triangle, utils.cpp:116-129; compute_barycentric_normals,
# stlreader.cpp:62-81; set_normals, utils.cpp:131-135).
#   A = (0,0,0) (1,0,0) (0,1,0): edge01 x edge02 = (0,0,1)
#   B = (0,0,0) (0,0,-1) (0,1,0): edge01 = (0,0,-1), edge02 = (0,1,0), cross = (1,0,0)
#   C = (5,5,2) (6,5,2) (5,6,2): normal (0,0,1), shares no vertex; plane parameter (2 + 2 + 2) / 3 = 2
# A and B share the vertices (0,0,0) and (0,1,0) bit for bit: their vertex normal there is
# normalize((0,0,1) + (1,0,0)) = (s, 0, s) with s = 1 / sqrt(2) rounded as binary32 does it (the squared
# length 2 is exact, one square root, one division per component); the other vertices keep their
# triangle's own normal.  base_normal = n0, u_normal = n1 - n0, v_normal = n2 - n0.
FOLD = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, -1, 0, 1, 0], [5, 5, 2, 6, 5, 2, 5, 6, 2]], np.float32)


def test_triangle_data_known_answers():
    s = np.float32(1.0) / np.sqrt(np.float32(2.0))
    one = np.float32(1.0)
    want = np.zeros((3, 30), np.float32)
    #            p0         p1          p2         edge01      edge02      normal     base_n     u_n               v_n        plane
    want[0, :28] = [0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, s, 0, s, -s, 0, one - s, 0, 0, 0, 0]
    want[1, :28] = [0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, -1, 0, 1, 0, 1, 0, 0, s, 0, s, one - s, 0, -s, 0, 0, 0, 0]
    want[2, :28] = [5, 5, 2, 6, 5, 2, 5, 6, 2, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 2]
    for who, make in builders():
        got = make(FOLD).triangle_data()
        assert np.array_equal(got, want), (who, got - want)
    # the same through a binary STL file (the file's own normals are ignored, stlreader.cpp:30-34)
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "fold.stl")
        pta.meshgen.write_stl(path, FOLD)
        assert np.array_equal(pta.HostScene(stl_path=path).triangle_data(), want)
        assert np.array_equal(orc.Scene(stl_path=path).triangle_data(), want)


# ---------------------------------------------------------------------------------------------------
# PassDescriptor::clamp_bounds (integrator.cpp:164-183) on a 10 x 6 frame, by hand:
#   (-3,-3, 5, 5): the start is clamped to (0,0), the size is NOT shortened by the 3 pixels that hung
#                  over (5 <= 10 - 0): columns 0..4, rows 0..4
#   (8, 4, 1000, 1000): size cut to what is left, 2 x 2: columns 8..9, rows 4..5
#   (20, 20, 3, 3): the start is clamped to the last pixel (9,5), size 1 x 1
#   (2, 1, -1, -1): -1 means the frame's size, then cut: columns 2..9, rows 1..5
# Each rectangle gets one sample; the pass counts below are the sums.
CLAMP_RECTS = [(-3, -3, 5, 5), (8, 4, 1000, 1000), (20, 20, 3, 3), (2, 1, -1, -1)]


def clamp_counts():
    want = np.zeros((6, 10), np.int32)
    want[0:5, 0:5] += 1
    want[4:6, 8:10] += 1
    want[5, 9] += 1
    want[1:6, 2:10] += 1
    return want


def test_clamp_bounds_known_answers_oracle():
    scene = orc.Scene(TEN)
    rgb = np.zeros((6, 10, 3), np.float32)
    passes = np.zeros((6, 10), np.int32)
    for k, r in enumerate(CLAMP_RECTS):
        scene.render(orc.cli_camera(), [], 10, 6, r, k, 1, seed=1, rgb=rgb, passes=passes)
    assert np.array_equal(passes, clamp_counts())


@pytest.mark.gpu
def test_clamp_bounds_known_answers_gpu():
    film = pta.Film(pta.GpuScene(pta.HostScene(TEN), 0), 10, 6)
    st = film.render(pta.cli_camera(), [], [r + (k, 1) for k, r in enumerate(CLAMP_RECTS)], seed=1)
    _, passes = film.read()
    assert np.array_equal(passes, clamp_counts())
    assert st.camera_samples == int(clamp_counts().sum())


# ---------------------------------------------------------------------------------------------------
# Work in the reference's visit order (what the bench line's algorithmic bytes are priced from): box
# tests = calls of kdTreeNode::ray_test (each begins with its own box, kdtree.cpp:273), triangle tests =
# iterations of the leaf loop (:281-292), counted by hand on TEN:
#   ray 1 of ORDER_RAYS (origin inside both extents): root, low (5 triangles), high (5)             -> 3, 10
#   ray 3 (near = high, hit at t = 1 > t_far = 1/6): root, high (5), then low (5)                    -> 3, 10
#   origin (4.875, -0.9, 2), direction (-0.25, 0, -2): near = high, Q is hit at t = 1 <= t_far =
#     (4 - 4.875) / -0.25 = 3.5, so the far child is never entered (:367)                            -> 2, 5
#   origin (4.8, 50, 2) straight down: near = high, its box is hit but none of its triangles; the far
#     child is entered because the near one missed (:393-394) and fails its own box test             -> 3, 5
#   origin (0, 0, 5) pointing up: the root's box is behind the ray                                   -> 1, 0
WORK_RAYS = [
    (ORDER_RAYS[0][0], ORDER_RAYS[0][1], (3, 10)),
    (ORDER_RAYS[2][0], ORDER_RAYS[2][1], (3, 10)),
    ((4.875, -0.9, 2.0), (-0.25, 0.0, -2.0), (2, 5)),
    ((4.8, 50.0, 2.0), (0.0, 0.0, -2.0), (3, 5)),
    ((0.0, 0.0, 5.0), (0.0, 0.0, 1.0), (1, 0)),
]


def test_reference_order_work_counts():
    scene = orc.Scene(TEN)
    for o, d, want in WORK_RAYS:
        got = scene.raycast_counts(np.array([o], np.float32), np.array([d], np.float32))
        assert got == want, (o, d, got, want)
    idx, t, _, _ = scene.raycast(np.array([WORK_RAYS[2][0]], np.float32), np.array([WORK_RAYS[2][1]], np.float32))
    assert (int(idx[0]), float(t[0])) == (IDX_Q, 1.0)
