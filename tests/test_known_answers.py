"""Known answers worked out BY HAND from the reference's source (Moller-Trumbore as written in
utils.cpp:138-163, the visibility test of integrator.cpp:127-129, Canvas::get_pixel of
canvas.cpp:39-45, the camera of integrator.cpp:197-209), on numbers that binary32 represents
exactly, so that every value below can be checked with pencil and paper.  They pin the oracle --
and, in the GPU half, the HIP path -- to the reference's conventions independently of either
implementation: which barycentric is u and which is v, that hits are two-sided, that the domain
boundaries u = 0 and u + v = 1 count as hits, that t is measured in units of the given direction,
that t <= 1e-8 is a miss, and that the light test compares the barycentric v with the distance."""
import numpy as np
import pytest

from conftest import orc, pta

# one triangle p0 = (0,0,0), p1 = (1,0,0), p2 = (0,1,0): edge01 = x, edge02 = y
TRI = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32)

# origin, direction, expected (index, t, u, v); index -1 = miss (t, u, v then 0)
CASES = [
    # from above, straight down: P = d x e2 = (1,0,0), det = 1, u = T.P = 0.25, Q = T x e1 = (0,1,-0.25),
    # v = d.Q = 0.25, t = e2.Q = 1
    ((0.25, 0.25, 1.0), (0, 0, -1), (0, 1.0, 0.25, 0.25)),
    # from below (det = -1): two-sided, t = 2
    ((0.25, 0.25, -2.0), (0, 0, 1), (0, 2.0, 0.25, 0.25)),
    # u + v = 1.5 > 1: miss
    ((0.75, 0.75, 1.0), (0, 0, -1), (-1, 0.0, 0.0, 0.0)),
    # u + v = 1 exactly: `u + v > 1` is false, a hit
    ((0.5, 0.5, 1.0), (0, 0, -1), (0, 1.0, 0.5, 0.5)),
    # u = 0 exactly (`u < 0` is false, a hit): o = (-0.25,0.5,1), d = (0.25,0,-1): P = (1,0,0.25), det = 1,
    # u = T.P = -0.25 + 0.25 = 0, Q = T x e1 = (0,1,-0.5), v = d.Q = 0.5, t = e2.Q = 1
    ((-0.25, 0.5, 1.0), (0.25, 0, -1), (0, 1.0, 0.0, 0.5)),
    # v = 0 exactly: o = (0.5,-0.25,1), d = (0,0.25,-1): P = (1,0,0), u = 0.5, Q = (0,1,0.25), v = 0.25 - 0.25 = 0
    ((0.5, -0.25, 1.0), (0, 0.25, -1), (0, 1.0, 0.5, 0.0)),
    # in the triangle's plane: det = 0, miss
    ((-1.0, 0.25, 0.0), (1, 0, 0), (-1, 0.0, 0.0, 0.0)),
    # origin on the triangle: t = 0 <= 1e-8, miss
    ((0.25, 0.25, 0.0), (0, 0, -1), (-1, 0.0, 0.0, 0.0)),
    # triangle behind the origin: t = -1, miss
    ((0.25, 0.25, -1.0), (0, 0, -1), (-1, 0.0, 0.0, 0.0)),
    # direction of length 2: det = 2, u and v unchanged, t = 0.5 (in units of the direction)
    ((0.25, 0.25, 1.0), (0, 0, -2), (0, 0.5, 0.25, 0.25)),
    # oblique, all dyadic: o = (0,0,1), d = (0.25,0.5,-1): P = d x e2 = (1,0,0.25), det = 1, u = T.P = 0.25,
    # Q = T x e1 = (0,1,0), v = d.Q = 0.5, t = e2.Q = 1
    ((0.0, 0.0, 1.0), (0.25, 0.5, -1), (0, 1.0, 0.25, 0.5)),
]


# Axis-parallel rays whose origin lies exactly on a face plane of the tree's root box are MISSED by
# the reference although the triangle test alone would hit: with direction x = 0 the reciprocal is
# inf, (box.min.x - o.x) * inf = 0 * inf = NaN, and the select forms of utils.h:29-35 turn that into
# entry = inf (real_min(NaN, inf) returns its second operand), so t_start = inf > t_end
# (utils.cpp:59-83).  Strict IEEE reproduces this; the tree walk and the GPU must too.
BOX_NAN_CASES = [
    ((0.0, 0.5, 1.0), (0, 0, -1), (0, 1.0, 0.0, 0.5)),      # what the triangle test alone gives
    ((0.5, 0.0, 4.0), (0, 0, -1), (0, 4.0, 0.5, 0.0)),
]


def _nan_arrays():
    o = np.array([c[0] for c in BOX_NAN_CASES], np.float32)
    d = np.array([c[1] for c in BOX_NAN_CASES], np.float32)
    alone = [np.array([c[2][k] for c in BOX_NAN_CASES], dt) for k, dt in enumerate((np.int32, np.float32, np.float32, np.float32))]
    miss = [np.full(len(o), -1, np.int32)] + [np.zeros(len(o), np.float32)] * 3
    return o, d, alone, miss


def _arrays():
    o = np.array([c[0] for c in CASES], np.float32)
    d = np.array([c[1] for c in CASES], np.float32)
    want = [np.array([c[2][k] for c in CASES], dt) for k, dt in enumerate((np.int32, np.float32, np.float32, np.float32))]
    return o, d, want


def _check(got, want, who):
    for g, w, name in zip(got, want, ("index", "t", "u", "v")):
        assert np.array_equal(np.asarray(g), w), (who, name, g, w)


def test_oracle_triangle_known_answers():
    o, d, want = _arrays()
    scene = orc.Scene(TRI)
    _check(scene.raycast(o, d), want, "oracle tree walk")
    _check(scene.raycast_brute(o, d), want, "oracle exhaustive search")
    no, nd, alone, miss = _nan_arrays()
    _check(scene.raycast_brute(no, nd), alone, "triangle test alone")
    _check(scene.raycast(no, nd), miss, "oracle tree walk, NaN in the root box test")
    # the light test of integrator.cpp:127-129 receives the closest hit's barycentric v where it
    # expects t: a light is visible iff nothing is hit or v > distance
    o1 = np.repeat(o[:1], 4, axis=0)
    d1 = np.repeat(d[:1], 4, axis=0)                     # the first case: v = 0.25
    dist = np.array([0.125, 0.25, 0.5, 7.0], np.float32)
    assert list(scene.shadow_visible(o1, d1, dist)) == [1, 0, 0, 0]
    assert list(scene.shadow_visible(o[2:3], d[2:3], dist[:1])) == [1]          # the missing ray: visible
    # ... and the flagged extension bounds the ray at the light instead: hit at t = 1
    assert list(scene.shadow_visible(o1, d1, np.array([0.5, 1.0, 1.5, 7.0], np.float32), bounded=True)) == [1, 1, 0, 0]


# Equal-t tie: A = TRI and B = (-1,0,0),(1,0,0),(-1,2,0) lie in the same plane; the ray of the first
# case hits both at t = 1 (for B: P = (2,0,0), det = 4, u = 2.5/4 = 0.625, Q = (0,2,-0.5), v = 0.5/4 =
# 0.125).  Both sit in one leaf, whose order is by box minimum in x (kdtree.cpp:34,68): B (-1) before
# A (0); the leaf loop keeps a hit only if it is strictly nearer (kdtree.cpp:286), so B wins whatever
# the order of the two in the mesh.
TIE_B = [-1, 0, 0, 1, 0, 0, -1, 2, 0]


def _tie_scene(b_first):
    rows = [TIE_B, TRI[0].tolist()] if b_first else [TRI[0].tolist(), TIE_B]
    return np.array(rows, np.float32), (0 if b_first else 1)


def test_oracle_equal_t_tie_goes_to_the_first_in_leaf_order():
    o = np.array([[0.25, 0.25, 1.0]], np.float32)
    d = np.array([[0, 0, -1]], np.float32)
    for b_first in (False, True):
        tris, b_index = _tie_scene(b_first)
        idx, t, u, v = orc.Scene(tris).raycast(o, d)
        assert (int(idx[0]), float(t[0]), float(u[0]), float(v[0])) == (b_index, 1.0, 0.625, 0.125)


# cast_ray without lights (integrator.cpp:51-156): a ray that leaves the scene returns the sky
# colour; one that hits the triangle from above is mirrored straight up (n = (0,0,1)), leaves the
# scene, and brings back 0.8 * sky (:107); with recursions = 0 a hit returns nothing at all.
SKY = (0.5, 0.25, 1.0)
CAST_O = np.array([[0.25, 0.25, 1.0], [0.75, 0.75, 1.0], [0.25, 0.25, 1.0]], np.float32)
CAST_D = np.array([[0, 0, -1], [0, 0, -1], [0, 0, -1]], np.float32)
CAST_WANT_10 = np.array([[np.float32(0.8) * np.float32(c) for c in SKY], list(SKY), [np.float32(0.8) * np.float32(c) for c in SKY]], np.float32)
CAST_WANT_0 = np.array([[0, 0, 0], list(SKY), [0, 0, 0]], np.float32)


def test_oracle_cast_ray_sky_known_answers():
    scene = orc.Scene(TRI)
    cam = orc.make_camera((0, 0, 0), (1, 0, 0), sky_color=SKY)
    got, counts = scene.cast_rays(cam, [], CAST_O, CAST_D, recursions=10)
    assert np.array_equal(got, CAST_WANT_10) and list(counts[:2]) == [5, 0]     # two bounce rays, no shadow rays
    got, counts = scene.cast_rays(cam, [], CAST_O, CAST_D, recursions=0)
    assert np.array_equal(got, CAST_WANT_0) and list(counts[:2]) == [3, 0]


def test_oracle_tonemap_known_answers():
    # Canvas::get_pixel: sum / max(passes, 1) * 255, clamped to [0, 255], truncated
    rgb = np.array([[3.0, 0.5, 8.0], [0.75, -1.0, 0.999], [0.25, 0.0, 1.0]], np.float32)
    passes = np.array([4, 1, 0], np.int32)
    out = orc.tonemap(rgb, passes)
    assert out.tolist() == [[191, 31, 255], [191, 0, 254], [63, 0, 255]]


def test_oracle_camera_known_answers():
    # get_ray_for_pixel (integrator.cpp:197-209) with camera at the origin looking along +x, up = +z:
    # right = (0,-1,0), up' = (0,0,1); W = H = 4, image plane width 1: the pixel (2,2) looks straight
    # ahead, (3,2) a quarter to the right, (2,1) a quarter up -- before normalisation
    cam = orc.make_camera((0, 0, 0), (1, 0, 0), (0, 0, 1), image_plane_width=1.0)
    o, d = orc.camera_rays(cam, 4, 4)
    d = d.reshape(4, 4, 3)
    assert not o.any()
    assert d[2, 2].tolist() == [1.0, 0.0, 0.0]

    def unit(v):
        v = np.array(v, np.float32)
        z = np.float32(v[0] * v[0] + np.float32(v[1] * v[1] + v[2] * v[2]))
        return (v / np.sqrt(z, dtype=np.float32)).tolist()
    assert d[2, 3].tolist() == unit((1, -0.25, 0))
    assert d[1, 2].tolist() == unit((1, 0, 0.25))
    assert d[0, 0].tolist() == unit((1, 0.5, 0.5))


@pytest.mark.gpu
def test_gpu_triangle_known_answers():
    o, d, want = _arrays()
    gpu = pta.GpuScene(pta.HostScene(TRI), 0)
    _check(gpu.raycast_closest(o, d), want, "GPU closest hit")
    no, nd, _, miss = _nan_arrays()
    _check(gpu.raycast_closest(no, nd), miss, "GPU closest hit, NaN in the root box test")
    o1 = np.repeat(o[:1], 4, axis=0)
    d1 = np.repeat(d[:1], 4, axis=0)
    assert list(gpu.raycast_shadow(o1, d1, np.array([0.125, 0.25, 0.5, 7.0], np.float32))) == [1, 0, 0, 0]
    assert list(gpu.raycast_shadow(o[2:3], d[2:3], np.array([0.125], np.float32))) == [1]
    assert list(gpu.raycast_shadow(o1, d1, np.array([0.5, 1.0, 1.5, 7.0], np.float32), bounded=True)) == [1, 1, 0, 0]


@pytest.mark.gpu
def test_gpu_equal_t_tie_goes_to_the_first_in_leaf_order():
    o = np.array([[0.25, 0.25, 1.0]], np.float32)
    d = np.array([[0, 0, -1]], np.float32)
    for b_first in (False, True):
        tris, b_index = _tie_scene(b_first)
        idx, t, u, v = pta.GpuScene(pta.HostScene(tris), 0).raycast_closest(o, d)
        assert (int(idx[0]), float(t[0]), float(u[0]), float(v[0])) == (b_index, 1.0, 0.625, 0.125)


@pytest.mark.gpu
def test_gpu_cast_ray_sky_known_answers():
    gpu = pta.GpuScene(pta.HostScene(TRI), 0)
    cam = pta.make_camera((0, 0, 0), (1, 0, 0), sky_color=SKY)
    got, st = gpu.cast_rays(cam, [], CAST_O, CAST_D, recursions=10)
    assert np.array_equal(got, CAST_WANT_10) and (st.closest_rays, st.shadow_rays) == (5, 0)
    got, st = gpu.cast_rays(cam, [], CAST_O, CAST_D, recursions=0)
    assert np.array_equal(got, CAST_WANT_0) and (st.closest_rays, st.shadow_rays) == (3, 0)
