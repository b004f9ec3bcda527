"""The light loop of cast_ray (reference integrator.cpp:112-148) against an INDEPENDENT evaluation of
its expectation, written here in numpy straight from the reference's lines: a point on a large floor,
seen from straight above, lit by the three lights of cli_render.cpp:89-91, each displaced by a
N(0, 0.2) vector per sample (:121-123); what a sample adds is colour / dist^2 * (max(0, n.l) +
max(0, r.l)^16) with l = to_light / dist (:131-145).  With n = r = (0,0,1) and nothing in the way the
mean over the displacement is a smooth 3-D Gaussian integral, evaluated below by Gauss-Hermite
quadrature; the renderer's mean over 2^17 independent samples must agree within its statistical error.
This pins the displacement's standard deviation, the 1/dist^2 and the 16th power, the sum over lights
and the absence of any other factor on direct light -- none of which a test against the oracle alone
could, since oracle and GPU share their formulation.

And the reference's visibility test as it stands (:127-129, kdtree.h:84): the shadow ray is not bounded
at the light, so a ceiling ABOVE every light hides them all -- every sample returns exactly nothing --
while the flagged bounded mode sees them."""
import numpy as np
import pytest

from conftest import orc, pta

FLOOR = [-10, -10, 0, 10, -10, 0, -10, 10, 0]                 # covers the point (-1, -2, 0): u = 0.45, v = 0.4
# a floor and a ceiling large enough that every shadow ray of the test meets one of them beyond its light
BIG_FLOOR = [-40, -40, 0, 120, -40, 0, -40, 120, 0]
CEILING = [-40, -40, 6, 120, -40, 6, -40, 120, 6]             # above every light (z = 3, 4, 4)
HIT = np.array([-1.0, -2.0, 1e-3])                             # the hit point lifted by 1e-3 along the normal (:72)
N_RAYS = 1 << 17


def expectation(hit=HIT, normal=(0.0, 0.0, 1.0), reflection=(0.0, 0.0, 1.0)):
    """Sum over the lights of E[colour / dist^2 * (max(0, n.l) + max(0, r.l)^16)], l = to_light / dist,
    to_light = displacement + light - hit, displacement ~ N(0, 0.2^2 I)."""
    x, w = np.polynomial.hermite_e.hermegauss(40)              # weight exp(-x^2 / 2)
    w = w / np.sqrt(2 * np.pi)
    zx, zy, zz = np.meshgrid(x, x, x, indexing="ij")
    ww = w[:, None, None] * w[None, :, None] * w[None, None, :]
    total = np.zeros(3)
    for pos, colour in orc.cli_lights():
        tl = [0.2 * z + p - h for z, p, h in zip((zx, zy, zz), pos, hit)]
        dist = np.sqrt(tl[0] ** 2 + tl[1] ** 2 + tl[2] ** 2)
        lambert = np.maximum(0.0, sum(n * t for n, t in zip(normal, tl)) / dist)
        phong = np.maximum(0.0, sum(r * t for r, t in zip(reflection, tl)) / dist)
        total += np.array(colour) * np.sum(ww * (lambert + phong ** 16) / dist ** 2)
    return total


# The folded pair of tests/test_known_answers_tree.py: A = (0,0,0) (1,0,0) (0,1,0) has the vertex normals
# (s,0,s), (0,0,1), (s,0,s), s = 1/sqrt(2) (its neighbour B hangs below the plane and is in no ray's way).
# A ray straight down onto (0.5, 0.25, 0) has u = 0.5, v = 0.25: the interpolated normal is
# normalize(n0 + u (n1 - n0) + v (n2 - n0)) = normalize((s/2, 0, s/2 + 1/2)) (integrator.cpp:66-67), the
# mirror direction normalize(d - 2 (n.d) n) (:73-74), the hit point is lifted along the GEOMETRIC normal (:72).
FOLD = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, -1, 0, 1, 0]], np.float32)


def fold_frame():
    s = 1.0 / np.sqrt(2.0)
    n = np.array([s / 2, 0.0, s / 2 + 0.5])
    n /= np.linalg.norm(n)
    d = np.array([0.0, 0.0, -1.0])
    r = d - 2 * n.dot(d) * n
    r /= np.linalg.norm(r)
    return np.array([0.5, 0.25, 1e-3]), n, r


def fold_rays():
    o = np.tile(np.array([[0.5, 0.25, 5.0]], np.float32), (N_RAYS, 1))
    d = np.tile(np.array([[0.0, 0.0, -1.0]], np.float32), (N_RAYS, 1))
    return o, d


def rays():
    o = np.tile(np.array([[-1.0, -2.0, 5.0]], np.float32), (N_RAYS, 1))
    d = np.tile(np.array([[0.0, 0.0, -1.0]], np.float32), (N_RAYS, 1))
    return o, d


def check_mean(rgb, who, want=None):
    want = expectation() if want is None else want
    got = rgb.astype(np.float64).mean(axis=0)
    err = rgb.astype(np.float64).std(axis=0) / np.sqrt(len(rgb))
    assert (np.abs(got - want) < 5 * err + 1e-4 * want).all(), (who, got, want, err)
    assert (err < 0.005 * want).all(), "the sample is large enough for the comparison to mean something"


def test_light_loop_expectation_oracle():
    o, d = rays()
    scene = orc.Scene(np.array([FLOOR], np.float32))
    rgb, counts = scene.cast_rays(orc.make_camera((0, 0, 0), (1, 0, 0)), orc.cli_lights(), o, d, recursions=10)
    assert list(counts[:2]) == [2 * N_RAYS, 3 * N_RAYS]       # one bounce ray that leaves, three shadow casts
    check_mean(rgb, "oracle")
    # interpolated normal and mirror direction
    o, d = fold_rays()
    rgb, counts = orc.Scene(FOLD).cast_rays(orc.make_camera((0, 0, 0), (1, 0, 0)), orc.cli_lights(), o, d, recursions=10)
    assert list(counts[:2]) == [2 * N_RAYS, 3 * N_RAYS]
    check_mean(rgb, "oracle, folded pair", expectation(*fold_frame()))
    flat = expectation(fold_frame()[0])
    assert (np.abs(expectation(*fold_frame()) - flat) > 0.02 * flat).all(), "the tilt of the normal matters at this point"


def test_unbounded_shadow_rays_oracle():
    o, d = rays()
    o, d = o[:4096], d[:4096]
    scene = orc.Scene(np.array([BIG_FLOOR, CEILING], np.float32))
    rgb, _ = scene.cast_rays(orc.make_camera((0, 0, 0), (1, 0, 0)), orc.cli_lights(), o, d, recursions=10)
    assert not rgb.any()
    rgb, _ = scene.cast_rays(orc.make_camera((0, 0, 0), (1, 0, 0), shadow_mode=1), orc.cli_lights(), o, d, recursions=10)
    assert (rgb > 0).all()


@pytest.mark.gpu
def test_light_loop_expectation_gpu():
    o, d = rays()
    gpu = pta.GpuScene(pta.HostScene(np.array([FLOOR], np.float32)), 0)
    rgb, st = gpu.cast_rays(pta.make_camera((0, 0, 0), (1, 0, 0)), pta.cli_lights(), o, d, recursions=10)
    assert (st.closest_rays, st.shadow_rays) == (2 * N_RAYS, 3 * N_RAYS)
    check_mean(rgb, "GPU")
    o2, d2 = fold_rays()
    rgb, _ = pta.GpuScene(pta.HostScene(FOLD), 0).cast_rays(pta.make_camera((0, 0, 0), (1, 0, 0)), pta.cli_lights(), o2, d2, recursions=10)
    check_mean(rgb, "GPU, folded pair", expectation(*fold_frame()))
    gpu = pta.GpuScene(pta.HostScene(np.array([BIG_FLOOR, CEILING], np.float32)), 0)
    rgb, _ = gpu.cast_rays(pta.make_camera((0, 0, 0), (1, 0, 0)), pta.cli_lights(), o[:4096], d[:4096], recursions=10)
    assert not rgb.any()
    rgb, _ = gpu.cast_rays(pta.make_camera((0, 0, 0), (1, 0, 0), shadow_mode=1), pta.cli_lights(), o[:4096], d[:4096], recursions=10)
    assert (rgb > 0).all()


# Two small facing patches, a floor piece (normal +z) and a ceiling piece (normal -z: the other winding),
# around the vertical through (-1, -2): a ray straight down bounces between them until its 10 recursions
# are used up (integrator.cpp:76, :107, :261) -- 11 path vertices, each lit by the three lights (no shadow
# ray meets a patch: they are half a unit wide and the lights are far off the vertical), the k-th weighted
# 0.8^k.  So the expectation is E_floor (1 + 0.8^2 + ... + 0.8^10) + E_ceiling (0.8 + 0.8^3 + ... + 0.8^9).
PATCHES = np.array([[-1.25, -2.25, 0, -0.25, -2.25, 0, -1.25, -1.25, 0],
                    [-1.25, -2.25, 6, -1.25, -1.25, 6, -0.25, -2.25, 6]], np.float32)


def bounce_series_expectation():
    e_floor = expectation(np.array([-1.0, -2.0, 1e-3]), (0.0, 0.0, 1.0), (0.0, 0.0, 1.0))
    e_ceiling = expectation(np.array([-1.0, -2.0, 6.0 - 1e-3]), (0.0, 0.0, -1.0), (0.0, 0.0, -1.0))
    return e_floor * sum(0.8 ** k for k in range(0, 11, 2)) + e_ceiling * sum(0.8 ** k for k in range(1, 11, 2))


def test_bounce_series_oracle():
    o, d = rays()
    rgb, counts = orc.Scene(PATCHES).cast_rays(orc.make_camera((0, 0, 0), (1, 0, 0)), orc.cli_lights(), o, d, recursions=10)
    assert list(counts[:2]) == [11 * N_RAYS, 33 * N_RAYS]
    check_mean(rgb, "oracle, bounce series", bounce_series_expectation())
    # fewer recursions, fewer terms: cast_ray(ray, 2, 1) stops after three vertices
    rgb, counts = orc.Scene(PATCHES).cast_rays(orc.make_camera((0, 0, 0), (1, 0, 0)), orc.cli_lights(), o, d, recursions=2)
    assert list(counts[:2]) == [3 * N_RAYS, 9 * N_RAYS]
    e_floor = expectation(np.array([-1.0, -2.0, 1e-3]))
    e_ceiling = expectation(np.array([-1.0, -2.0, 6.0 - 1e-3]), (0.0, 0.0, -1.0), (0.0, 0.0, -1.0))
    check_mean(rgb, "oracle, three vertices", e_floor * (1 + 0.64) + e_ceiling * 0.8)


@pytest.mark.gpu
def test_bounce_series_gpu():
    o, d = rays()
    gpu = pta.GpuScene(pta.HostScene(PATCHES), 0)
    rgb, st = gpu.cast_rays(pta.make_camera((0, 0, 0), (1, 0, 0)), pta.cli_lights(), o, d, recursions=10)
    assert (st.closest_rays, st.shadow_rays) == (11 * N_RAYS, 33 * N_RAYS)
    check_mean(rgb, "GPU, bounce series", bounce_series_expectation())
