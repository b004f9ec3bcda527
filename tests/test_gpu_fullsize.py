"""Parity at the full size of the ray-casting configuration (1.34 M triangles, the 2 073 600
pixel-centre primaries of the default camera, 2^20 incoherent rays): bit-exact against the CPU
oracle on a sample the oracle finishes in seconds, and on the whole sets through properties that
do not depend on size -- results do not depend on the order or the batching of the rays, any-hit
visibility agrees with the closest hit, and every reported hit is a valid one."""
import numpy as np
import pytest

from conftest import assert_hits_equal, bits, orc, pta, rays_sphere

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    tris = pta.meshgen.standin_scene(8)
    host = pta.HostScene(tris, threads=16)
    return tris, host, pta.GpuScene(host, 0)


def _primaries():
    o, d = orc.camera_rays(orc.cli_camera(), 1920, 1080)
    return np.ascontiguousarray(o.reshape(-1, 3)), np.ascontiguousarray(d.reshape(-1, 3))


def test_sample_against_oracle(big):
    tris, host, gpu = big
    assert host.n == 1343488
    oracle = orc.Scene(tris, threads=16)
    rng = np.random.default_rng(2)
    po, pd = _primaries()
    so, sd = rays_sphere(1 << 20)
    for name, (o, d) in (("primary", (po, pd)), ("sphere", (so, sd))):
        got = gpu.raycast_closest(o, d)
        sel = rng.choice(len(o), 60000, replace=False)
        want = oracle.raycast(np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel]), threads=16)
        assert_hits_equal([g[sel] for g in got], want, name)
        idx, t, u, v = got
        hit = idx >= 0
        assert 0.3 < hit.mean() < 1.0
        assert (t[hit] > 0).all() and (u[hit] >= 0).all() and (v[hit] >= 0).all() and (u[hit] + v[hit] <= 1).all()
        assert (idx[hit] < host.n).all()
        assert not t[~hit].any() and not u[~hit].any() and not v[~hit].any()


def test_order_and_batching_do_not_matter(big):
    _, _, gpu = big
    o, d = _primaries()
    whole = gpu.raycast_closest(o, d)
    perm = np.random.default_rng(3).permutation(len(o))
    shuffled = gpu.raycast_closest(np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm]))
    assert_hits_equal(shuffled, [w[perm] for w in whole], "shuffled")
    # ragged pieces, concatenated
    cuts = [0, 1, 64, 65, 100001, 1 << 20, len(o)]
    pieces = [gpu.raycast_closest(np.ascontiguousarray(o[a:b]), np.ascontiguousarray(d[a:b])) for a, b in zip(cuts, cuts[1:])]
    assert_hits_equal([np.concatenate([p[k] for p in pieces]) for k in range(4)], whole, "pieces")
    # a second run gives the same bits (no dependence on scheduling)
    again = gpu.raycast_closest(o, d)
    assert all(np.array_equal(bits(a), bits(b)) for a, b in zip(again, whole))


def test_any_hit_agrees_with_closest_hit(big):
    _, _, gpu = big
    o, d = rays_sphere(1 << 20, seed=77)
    idx, t, u, v = gpu.raycast_closest(o, d)
    far = np.full(len(o), 2.0, np.float32)           # light at distance >= 1: any hit occludes
    assert np.array_equal(gpu.raycast_shadow(o, d, far) == 0, idx >= 0)
    near = np.random.default_rng(5).random(len(o)).astype(np.float32)   # < 1: needs the closest hit's v
    assert np.array_equal(gpu.raycast_shadow(o, d, near) == 1, (idx < 0) | (v > near))


def test_full_frame_render(big):
    """The path-tracing configuration at full size (1920x1080, DOF 0.1/4.5, 3 lights), 2 samples:
    a band of the frame against the oracle (identical ray counts, radiance within a few ulp:
    bounce terms are summed in another order), and the whole frame tiled 64x64 as the reference
    renders it against one full-frame pass, bit for bit."""
    tris, host, gpu = big
    W, H, S = 1920, 1080, 2
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    film = pta.Film(gpu, W, H)
    st = film.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=9)
    full, passes = film.read()
    assert (passes == S).all() and st.camera_samples == W * H * S
    assert st.closest_rays >= st.camera_samples and st.shadow_rays % 3 == 0

    film.zero()
    tiles = [(x, y, 64, 64, s, 1) for s in range(S) for y in range(0, H, 64) for x in range(0, W, 64)]
    st2 = film.render(cam, pta.cli_lights(), tiles, seed=9)
    tiled, passes2 = film.read()
    assert (passes2 == S).all()
    assert (st2.closest_rays, st2.shadow_rays) == (st.closest_rays, st.shadow_rays)
    assert np.array_equal(full.view(np.uint32), tiled.view(np.uint32))

    oracle = orc.Scene(tris, threads=16)
    band = (0, 536, W, 8)
    want, want_p, counts = oracle.render(orc.cli_camera(dof_aperture=0.1, dof_distance=4.5), orc.cli_lights(), W, H, band, 0, S,
                                         seed=9, threads=16)
    film.zero()
    st3 = film.render(cam, pta.cli_lights(), [band + (0, S)], seed=9)
    got, got_p = film.read()
    assert np.array_equal(got_p, want_p)
    assert (st3.closest_rays, st3.shadow_rays) == (int(counts[0]), int(counts[1]))
    assert st3.shadow_rays_traced == int(counts[2])
    rows = slice(536, 544)
    assert np.array_equal(got[rows].view(np.uint32), full[rows].view(np.uint32))   # a band is part of the frame
    err = np.abs(got - want)
    assert (err <= 1e-5 * np.maximum(np.abs(want), 1.0)).all(), err.max()


def test_full_frame_lambert_band(big):
    """The flagged Lambertian bounce mode (integrator.cpp:79-84 without :104, the README's "diffuse"
    look) at the full size of the path-tracing configuration: an 8-row band of the 1920x1080 frame,
    2 samples, against the oracle -- identical ray counts (bounce directions are bit-exact, so every
    path meets the same triangles), radiance within the summation-order tolerance -- and a band
    rendered alone equals the same rows of a full-frame render."""
    tris, host, gpu = big
    W, H, S = 1920, 1080, 2
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5, bounce_mode=1)
    band = (0, 536, W, 8)
    oracle = orc.Scene(tris, threads=16)
    want, want_p, counts = oracle.render(orc.cli_camera(dof_aperture=0.1, dof_distance=4.5, bounce_mode=1), orc.cli_lights(), W, H,
                                         band, 0, S, seed=9, threads=16)
    film = pta.Film(gpu, W, H)
    st = film.render(cam, pta.cli_lights(), [band + (0, S)], seed=9)
    got, got_p = film.read()
    assert np.array_equal(got_p, want_p)
    assert (st.closest_rays, st.shadow_rays, st.shadow_rays_traced) == (int(counts[0]), int(counts[1]), int(counts[2]))
    err = np.abs(got - want)
    assert (err <= 1e-5 * np.maximum(np.abs(want), 1.0)).all(), err.max()
    # diffuse paths live longer than mirror paths on this scene: more casts per camera sample
    mirror = film.render(pta.cli_camera(dof_aperture=0.1, dof_distance=4.5), pta.cli_lights(), [band + (S, S)], seed=9)
    assert st.closest_rays > mirror.closest_rays
    film.zero()
    film.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=9)
    full, _ = film.read()
    rows = slice(536, 544)
    assert np.array_equal(got[rows].view(np.uint32), full[rows].view(np.uint32))
