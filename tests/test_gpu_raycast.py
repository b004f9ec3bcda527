"""GPU parity of the ray-casting entry points, through the C ABI, against the CPU oracle.
Bar: bit-exact triangle index, t, u, v (integer/bit equality, no tolerance)."""
import numpy as np
import pytest

from conftest import all_ray_sets, assert_hits_equal, orc, pta, rays_sphere

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which", ["scene_tiny", "scene_small", "scene_medium"])
def test_closest_hit_bit_exact(which, request):
    sc = request.getfixturevalue(which)
    for name, (o, d) in all_ray_sets(sc.tris).items():
        want = sc.oracle.raycast(o, d)
        got = sc.gpu.raycast_closest(o, d)
        assert_hits_equal(got, want, f"{which}/{name}")


def test_closest_hit_ragged_sizes(scene_small):
    sc = scene_small
    o, d = all_ray_sets(sc.tris, 3000)["sphere"]
    for n in (1, 63, 64, 65, 257, 2999):
        assert_hits_equal(sc.gpu.raycast_closest(o[:n], d[:n]), sc.oracle.raycast(o[:n], d[:n]), f"n={n}")
    # empty input is a no-op
    got = sc.gpu.raycast_closest(o[:0], d[:0])
    assert all(x.size == 0 for x in got)


def test_shadow_visibility(scene_small, scene_medium):
    for sc in (scene_small, scene_medium):
        for name, (o, d) in all_ray_sets(sc.tris, 8000).items():
            rng = np.random.default_rng(11)
            # light distances on both sides of 1: below 1 the reference's swapped out-parameters
            # make the answer depend on the closest hit's barycentric v
            dist = np.where(rng.random(len(o)) < 0.5, rng.random(len(o)), 1.0 + 4.0 * rng.random(len(o))).astype(np.float32)
            want = sc.oracle.shadow_visible(o, d, dist)
            got = sc.gpu.raycast_shadow(o, d, dist)
            assert np.array_equal(got, want), f"{name}: {(got != want).sum()} differ"


def test_empty_scene_misses():
    host = pta.HostScene(np.zeros((0, 9), np.float32))
    gpu = pta.GpuScene(host, 0)
    o = np.zeros((100, 3), np.float32)
    d = np.tile(np.array([[0, 0, 1]], np.float32), (100, 1))
    idx, t, u, v = gpu.raycast_closest(o, d)
    assert (idx == -1).all() and (t == 0).all()
    assert (gpu.raycast_shadow(o, d, np.full(100, 2.0, np.float32)) == 1).all()


def test_camera_rays_bit_exact(scene_small):
    import torch
    sc = scene_small
    W, H = 96, 64
    for dof in (0.0, 0.1):
        cam_o = orc.cli_camera(dof_aperture=dof, dof_distance=4.5)
        cam_g = pta.cli_camera(dof_aperture=dof, dof_distance=4.5)
        for jittered, sample in ((0, 0), (1, 0), (1, 7)):
            out = [torch.empty(W * H, dtype=torch.float32, device="cuda") for _ in range(6)]
            sc.gpu.camera_rays_dev(cam_g, W, H, (0, 0, W, H), jittered, sample, 99, out)
            sc.gpu.sync()
            got = np.stack([t.cpu().numpy() for t in out], axis=1)
            o, d = orc.camera_rays(cam_o, W, H, None, bool(jittered), sample, 99)
            want = np.concatenate([o, d], axis=1)
            bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0]
            assert bad.size == 0, f"dof={dof} jittered={jittered}: {bad.size} rays differ, e.g. {got[bad[:2]]} vs {want[bad[:2]]}"


def test_render_matches_oracle(scene_small):
    sc = scene_small
    W, H, S = 96, 64, 4
    cam_o = orc.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    cam_g = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    want_rgb, want_p, want_counts = sc.oracle.render(cam_o, orc.cli_lights(), W, H, None, 0, S, seed=3)
    film = pta.Film(sc.gpu, W, H)
    st = film.render(cam_g, pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=3)
    rgb, passes = film.read()
    assert np.array_equal(passes, want_p)
    assert st.camera_samples == W * H * S
    assert st.closest_rays == int(want_counts[0]) and st.shadow_rays == int(want_counts[1])
    assert st.shadow_rays_traced == int(want_counts[2])   # the light samples with a non-zero weight
    # Paths are identical ray for ray (same hits, same random stream); only the order in which
    # the bounce terms are summed differs (0.8*(0.8*x) nested in the oracle, a running
    # throughput on the GPU), so the tolerance is a few ulp of the pixel value.
    err = np.abs(rgb - want_rgb)
    tol = 1e-5 * np.maximum(np.abs(want_rgb), 1.0)
    assert (err <= tol).all(), f"max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"
    rmse = float(np.sqrt(np.mean((rgb / S - want_rgb / S) ** 2)))
    assert rmse < 1e-6, rmse


def test_render_tiling_invariance(scene_small):
    """The image does not depend on how passes are cut into tiles or batches."""
    sc = scene_small
    W, H, S = 80, 48, 3
    cam = pta.cli_camera(dof_aperture=0.05, dof_distance=4.5)
    film = pta.Film(sc.gpu, W, H)
    film.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=5)
    full, _ = film.read()
    film.zero()
    tiles = []
    for s in range(S):
        for y in range(0, H, 32):
            for x in range(0, W, 32):
                tiles.append((x, y, 32, 32, s, 1))
    film.render(cam, pta.cli_lights(), tiles, seed=5)
    tiled, passes = film.read()
    assert (passes == S).all()
    assert np.array_equal(full.view(np.uint32), tiled.view(np.uint32))


def test_film_add_restores_accumulators(scene_tiny):
    """pt_film_add: a film that is read, zeroed and given its accumulators back is unchanged."""
    gpu = pta.GpuScene(scene_tiny.host, 0)
    film = pta.Film(gpu, 64, 48)
    film.render(pta.cli_camera(), pta.cli_lights(), [(0, 0, -1, -1, 0, 3)], seed=5)
    rgb, passes = film.read()
    assert (passes == 3).all() and rgb.max() > 0
    film.zero()
    film.add(rgb, passes)
    film.add(rgb, passes)
    rgb2, passes2 = film.read()
    assert np.array_equal(passes2, 2 * passes) and np.array_equal(rgb2, rgb + rgb)
    with pytest.raises(ValueError):
        film.add(rgb[:10], passes[:10])


def test_lambertian_render_matches_oracle(scene_small):
    """The flagged Lambertian bounce mode: identical ray counts and pass counts, radiance within a
    few ulp of the oracle's (same paths; bounce terms summed in another order)."""
    sc = scene_small
    W, H, S = 96, 64, 3
    want_rgb, want_p, want_counts = sc.oracle.render(orc.cli_camera(dof_aperture=0.1, dof_distance=4.5, bounce_mode=1),
                                                     orc.cli_lights(), W, H, None, 0, S, seed=6)
    film = pta.Film(sc.gpu, W, H)
    st = film.render(pta.cli_camera(dof_aperture=0.1, dof_distance=4.5, bounce_mode=1), pta.cli_lights(),
                     [(0, 0, -1, -1, 0, S)], seed=6)
    rgb, passes = film.read()
    assert np.array_equal(passes, want_p)
    assert st.closest_rays == int(want_counts[0]) and st.shadow_rays == int(want_counts[1])
    err = np.abs(rgb - want_rgb)
    assert (err <= 1e-5 * np.maximum(np.abs(want_rgb), 1.0)).all(), err.max()
    film.zero()
    film.render(pta.cli_camera(dof_aperture=0.1, dof_distance=4.5), pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=6)
    mirror, _ = film.read()
    assert not np.array_equal(mirror, rgb)
    with pytest.raises(RuntimeError):
        film.render(pta.cli_camera(bounce_mode=7), pta.cli_lights(), [(0, 0, -1, -1, 0, 1)], seed=6)


@pytest.mark.parametrize("shadow_mode", [0, 1])
def test_shadow_traces_beside_the_next_closest_hit_trace(scene_medium, monkeypatch, shadow_mode):
    """PT_OVERLAP_SHADOW=1 runs the shadow traces of depth d on a second stream beside the closest-hit
    trace of depth d+1 (an option).  Same film bit for bit, same ray counts, several batches in a row,
    in both shadow modes, and against the oracle."""
    sc = scene_medium
    W, H, S = 256, 128, 6
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5, shadow_mode=shadow_mode)
    film = pta.Film(sc.gpu, W, H)
    monkeypatch.setenv("PT_BATCH_PATHS", str(W * H * 2))   # three batches per call
    st1 = film.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=5)
    plain, p1 = film.read()
    film.zero()
    monkeypatch.setenv("PT_OVERLAP_SHADOW", "1")
    st2 = film.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=5)
    beside, p2 = film.read()
    assert (p1 == S).all() and np.array_equal(p1, p2)
    assert (st1.closest_rays, st1.shadow_rays, st1.shadow_rays_traced) == (st2.closest_rays, st2.shadow_rays, st2.shadow_rays_traced)
    assert np.array_equal(plain.view(np.uint32), beside.view(np.uint32))
    want_rgb, want_p, counts = sc.oracle.render(orc.cli_camera(dof_aperture=0.1, dof_distance=4.5, shadow_mode=shadow_mode), orc.cli_lights(),
                                                W, H, None, 0, S, seed=5)
    assert st2.closest_rays == int(counts[0]) and st2.shadow_rays == int(counts[1])
    assert (np.abs(beside - want_rgb) <= 1e-5 * np.maximum(np.abs(want_rgb), 1.0)).all()


def test_two_render_lanes_match_one(scene_small, monkeypatch):
    """PT_RENDER_LANES=2 keeps two batches in flight on two streams.  Cut into the same four
    batches, the two-lane render must leave the very same film and ray counts as the one-lane
    render: batches reach the film in order whichever stream they ran on."""
    sc = scene_small
    W, H, S = 1024, 512, 32  # 2^19 paths per sample, 2^24 in all
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    film = pta.Film(sc.gpu, W, H)
    monkeypatch.setenv("PT_RENDER_LANES", "1")
    monkeypatch.setenv("PT_BATCH_PATHS", str(1 << 22))
    st1 = film.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=11)
    one, p1 = film.read()
    film.zero()
    monkeypatch.setenv("PT_RENDER_LANES", "2")
    monkeypatch.setenv("PT_BATCH_PATHS", str(1 << 23))
    st2 = film.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=11)
    two, p2 = film.read()
    assert (p1 == S).all() and np.array_equal(p1, p2)
    assert (st1.camera_samples, st1.closest_rays, st1.shadow_rays) == (st2.camera_samples, st2.closest_rays, st2.shadow_rays)
    assert st1.camera_samples == W * H * S
    assert np.array_equal(one.view(np.uint32), two.view(np.uint32))
    # and a second two-lane call on the same film reuses both lanes' scratch
    film.render(cam, pta.cli_lights(), [(0, 0, -1, -1, S, S)], seed=11)
    _, p3 = film.read()
    assert (p3 == 2 * S).all()


def test_oversized_passes_are_cut_into_bands(scene_small, monkeypatch):
    """A pass whose samples exceed the batch allowance is rendered as horizontal bands that each carry
    all of its samples (bands too large even at eight rows are cut by samples as well).  Whatever the
    cut, the film and the ray counts are those of the uncut pass, bit for bit: a pixel's samples reach
    the film in sample order.  Odd sizes and an offset rectangle included."""
    sc = scene_small
    W, H, S = 500, 301, 24
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    film = pta.Film(sc.gpu, W, H)
    passes = [(0, 0, -1, -1, 0, S), (37, 50, 333, 173, S, 7)]
    monkeypatch.setenv("PT_RENDER_LANES", "1")
    monkeypatch.setenv("PT_BATCH_PATHS", str(1 << 26))          # everything in one batch
    want_st = film.render(cam, pta.cli_lights(), passes, seed=5)
    want, want_p = film.read()
    assert want_st.camera_samples == W * H * S + 333 * 173 * 7
    per_row = ((W + 7) // 8) * 64                                # one row of 8x8 blocks, one sample
    for allowance in (per_row * S * 5 + 17,                       # bands of five block rows
                      per_row * S + 1,                            # bands of one block row
                      per_row * 5):                               # one block row holds only five samples
        film.zero()
        monkeypatch.setenv("PT_BATCH_PATHS", str(max(allowance, 4096)))
        st = film.render(cam, pta.cli_lights(), passes, seed=5)
        got, got_p = film.read()
        assert np.array_equal(got_p, want_p), allowance
        assert (st.camera_samples, st.closest_rays, st.shadow_rays, st.shadow_rays_traced) == \
               (want_st.camera_samples, want_st.closest_rays, want_st.shadow_rays, want_st.shadow_rays_traced), allowance
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), allowance


def test_bounded_occlusion(scene_small, scene_medium):
    """pt_raycast_occluded (a flagged extension, no counterpart in the reference): visible iff
    no triangle in any node whose box the ray hits is hit at t < max_t -- the oracle's
    order-independent search.  Bounds are drawn around each ray's own closest-hit distance, equal
    to it (strict '<': visible unless another triangle is hit a hair nearer), and NaN/inf."""
    seen = [0, 0]
    for sc in (scene_small, scene_medium):
        for name, (o, d) in all_ray_sets(sc.tris, 8000).items():
            idx, t, _, _ = sc.oracle.raycast(o, d)
            rng = np.random.default_rng(12)
            k = rng.integers(0, 6, len(o))
            base = np.where(idx >= 0, t, np.float32(1.0)).astype(np.float32)
            bound = np.select([k == 0, k == 1, k == 2, k == 3, k == 4],
                              [base, np.nextafter(base, np.float32(np.inf)), base * np.float32(0.5), base * np.float32(3.0),
                               np.full(len(o), np.inf, np.float32)], np.full(len(o), np.nan, np.float32)).astype(np.float32)
            want = sc.oracle.shadow_visible(o, d, bound, bounded=True)
            with np.errstate(invalid="ignore"):
                by_closest = (~((idx >= 0) & (t < bound))).astype(np.uint8)
            # the closest hit gives the same answer except on near-ties at shared vertices and edges
            assert (want != by_closest).mean() < 1e-3
            got = sc.gpu.raycast_shadow(o, d, bound, bounded=True)
            assert np.array_equal(got, want), f"{name}: {(got != want).sum()} differ"
            seen[0] += int(want.sum()); seen[1] += int(len(want) - want.sum())
    assert seen[0] > 1000 and seen[1] > 1000


def test_bounded_shadow_render_matches_oracle(scene_small):
    """shadow_mode 1: same ray counts as the reference mode (one shadow ray per light sample),
    radiance within a few ulp of the oracle's.  A fourth light low beside the object makes the
    modes differ: seen from the floor behind it, the unbounded shadow ray goes on into the object."""
    sc = scene_small
    W, H, S = 96, 64, 3
    kw = dict(dof_aperture=0.1, dof_distance=4.5)
    lights = pta.cli_lights() + [((-2.5, -1.0, -0.35), (9.0, 9.0, 9.0))]
    want_rgb, want_p, want_counts = sc.oracle.render(orc.cli_camera(shadow_mode=1, **kw), lights, W, H, None, 0, S, seed=8)
    film = pta.Film(sc.gpu, W, H)
    st = film.render(pta.cli_camera(shadow_mode=1, **kw), lights, [(0, 0, -1, -1, 0, S)], seed=8)
    rgb, passes = film.read()
    assert np.array_equal(passes, want_p)
    assert st.closest_rays == int(want_counts[0]) and st.shadow_rays == int(want_counts[1])
    err = np.abs(rgb - want_rgb)
    assert (err <= 1e-5 * np.maximum(np.abs(want_rgb), 1.0)).all(), err.max()
    film.zero()
    st0 = film.render(pta.cli_camera(**kw), lights, [(0, 0, -1, -1, 0, S)], seed=8)
    ref_rgb, _ = film.read()
    assert (st0.closest_rays, st0.shadow_rays) == (st.closest_rays, st.shadow_rays)
    want_ref, _, _ = sc.oracle.render(orc.cli_camera(**kw), lights, W, H, None, 0, S, seed=8)
    assert (np.abs(ref_rgb - want_ref) <= 1e-5 * np.maximum(np.abs(want_ref), 1.0)).all()
    assert np.abs(ref_rgb - rgb).max() > 1e-2
    with pytest.raises(RuntimeError):
        film.render(pta.cli_camera(shadow_mode=2), pta.cli_lights(), [(0, 0, -1, -1, 0, 1)], seed=8)


def test_zero_weight_shadow_rays_are_counted_not_traced(scene_small, monkeypatch):
    """A light sample whose weight is exactly zero adds nothing whether its light is visible or
    not, so its shadow ray is counted (the reference casts it) but not traced.  With
    PT_TRACE_ALL_SHADOW=1 every shadow ray is traced: same film bit for bit, same reference counts."""
    sc = scene_small
    W, H, S = 128, 96, 3
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    film = pta.Film(sc.gpu, W, H)
    monkeypatch.setenv("PT_TRACE_ALL_SHADOW", "1")
    st_all = film.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=17)
    rgb_all, p_all = film.read()
    film.zero()
    monkeypatch.delenv("PT_TRACE_ALL_SHADOW")
    st = film.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=17)
    rgb, p = film.read()
    assert np.array_equal(rgb.view(np.uint32), rgb_all.view(np.uint32)) and np.array_equal(p, p_all)
    assert (st.closest_rays, st.shadow_rays) == (st_all.closest_rays, st_all.shadow_rays)
    assert st_all.shadow_rays_traced == st_all.shadow_rays
    assert 0 < st.shadow_rays_traced < st.shadow_rays


def test_cast_rays_matches_oracle(scene_medium):
    """pt_cast_rays = Integrator::cast_ray for the caller's own rays: radiance per ray against the
    oracle's recursion for full depth, three bounces and none, sky and both extensions included."""
    sc = scene_medium
    cam_o = orc.cli_camera()
    o, d = orc.camera_rays(cam_o, 96, 64)
    so, sd = rays_sphere(4000, seed=31)
    o = np.concatenate([o, so]); d = np.concatenate([d, sd])
    ids = (np.arange(len(o), dtype=np.uint32) * np.uint32(2654435761)) | np.uint32(0x80000000)
    for rec, sky, bounce, shadow in ((10, (0, 0, 0), 0, 0), (3, (0.2, 0.3, 0.4), 0, 0), (0, (0, 0, 0), 0, 0), (10, (0.1, 0.1, 0.1), 1, 1)):
        co = orc.cli_camera(bounce_mode=bounce, shadow_mode=shadow)
        cg = pta.cli_camera(bounce_mode=bounce, shadow_mode=shadow)
        co.sky_color[:] = list(sky)
        cg.sky_color[:] = list(sky)
        want, counts = sc.oracle.cast_rays(co, orc.cli_lights(), o, d, ids=ids, sample=7, recursions=rec, seed=12)
        got, st = sc.gpu.cast_rays(cg, pta.cli_lights(), o, d, stream_ids=ids, sample=7, recursions=rec, seed=12)
        assert (st.closest_rays, st.shadow_rays) == (int(counts[0]), int(counts[1])), rec
        assert (np.abs(got - want) <= 1e-5 * np.maximum(np.abs(want), 1.0)).all(), (rec, np.abs(got - want).max())
        assert got.sum() > 0
    # branches (integrator.cpp:77, :107): the first hit's bounce term as the mean of several sub-paths
    for rec, branches, bounce in ((10, 4, 0), (3, 3, 1), (10, 0, 0), (0, 5, 0)):
        co = orc.cli_camera(bounce_mode=bounce)
        cg = pta.cli_camera(bounce_mode=bounce)
        co.sky_color[:] = [0.05, 0.1, 0.2]
        cg.sky_color[:] = [0.05, 0.1, 0.2]
        want, counts = sc.oracle.cast_rays(co, orc.cli_lights(), o, d, ids=ids, sample=2, recursions=rec, seed=5, branches=branches)
        got, st = sc.gpu.cast_rays(cg, pta.cli_lights(), o, d, stream_ids=ids, sample=2, recursions=rec, seed=5, branches=branches)
        assert (st.closest_rays, st.shadow_rays, st.shadow_rays_traced) == (int(counts[0]), int(counts[1]), int(counts[2])), (rec, branches)
        assert (np.abs(got - want) <= 1e-5 * np.maximum(np.abs(want), 1.0)).all(), (rec, branches, np.abs(got - want).max())
        one, st1 = sc.gpu.cast_rays(cg, pta.cli_lights(), o, d, stream_ids=ids, sample=2, recursions=rec, seed=5, branches=1)
        if rec > 0 and branches > 1:
            assert st.closest_rays > st1.closest_rays             # more sub-paths, more casts
            if bounce == 1:
                assert not np.allclose(got, one)                  # independent scatter directions: another estimate
        if branches <= 0:                                          # the loop at :77 does not run: lights of the first hit only
            none, _ = sc.gpu.cast_rays(cg, pta.cli_lights(), o, d, stream_ids=ids, sample=2, recursions=0, seed=5)
            assert np.array_equal(got, none)
    with pytest.raises(Exception):
        sc.gpu.cast_rays(pta.cli_camera(), pta.cli_lights(), o[:4], d[:4], branches=17)
    # without ids the ray's number keys the stream
    want, _ = sc.oracle.cast_rays(cam_o, orc.cli_lights(), o[:500], d[:500], sample=1, recursions=10, seed=3)
    got, _ = sc.gpu.cast_rays(pta.cli_camera(), pta.cli_lights(), o[:500], d[:500], sample=1, recursions=10, seed=3)
    assert (np.abs(got - want) <= 1e-5 * np.maximum(np.abs(want), 1.0)).all()
    with pytest.raises(Exception):
        sc.gpu.cast_rays(pta.cli_camera(), pta.cli_lights(), o[:4], d[:4], recursions=11)


@pytest.mark.parametrize("per_round", ["2", "3"])
def test_several_triangles_per_round_kernels(scene_small, scene_medium, monkeypatch, per_round):
    """Scenes whose leaves hold sixteen and more triangles on average get trace kernels that test
    two or three triangles of a leaf per round; PT_TRIS_PER_ROUND selects them for any scene.  Same
    results, bit for bit, on every ray set, for closest hit, both shadow tests and a render."""
    monkeypatch.setenv("PT_TRIS_PER_ROUND", per_round)
    for sc in (scene_small, scene_medium):
        gpu = pta.GpuScene(sc.host, 0)
        for name, (o, d) in all_ray_sets(sc.tris, 6000).items():
            want = sc.oracle.raycast(o, d)
            assert_hits_equal(gpu.raycast_closest(o, d), want, f"{per_round} per round/{name}")
            rng = np.random.default_rng(11)
            dist = np.where(rng.random(len(o)) < 0.5, rng.random(len(o)), 1.0 + 4.0 * rng.random(len(o))).astype(np.float32)
            assert np.array_equal(gpu.raycast_shadow(o, d, dist), sc.oracle.shadow_visible(o, d, dist)), name
            bound = np.where(want[0] >= 0, want[1] * np.float32(1.5), np.float32(1.0)).astype(np.float32)
            assert np.array_equal(gpu.raycast_shadow(o, d, bound, bounded=True), sc.oracle.shadow_visible(o, d, bound, bounded=True)), name
    W, H, S = 64, 48, 3
    film_a, film_b = pta.Film(gpu, W, H), pta.Film(scene_medium.gpu, W, H)
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    st_a = film_a.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=4)
    st_b = film_b.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, S)], seed=4)
    assert (st_a.closest_rays, st_a.shadow_rays) == (st_b.closest_rays, st_b.shadow_rays)
    assert np.array_equal(film_a.read()[0].view(np.uint32), film_b.read()[0].view(np.uint32))


def test_any_hit_walks_with_and_without_quantised_nodes(scene_medium, monkeypatch):
    """With PT_QNODES=1 the any-hit kernels read interior nodes whose children are both interior as
    32-byte quantised records (an option: it saves load bytes and costs instructions).  Same visibility
    bit for bit, for both shadow tests, and the same film."""
    sc = scene_medium
    monkeypatch.setenv("PT_QNODES", "1")
    plain = pta.GpuScene(sc.host, 0)       # "plain" in name only: this one walks q-nodes, sc.gpu does not
    monkeypatch.delenv("PT_QNODES")
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    films = []
    for g in (sc.gpu, plain):
        f = pta.Film(g, 200, 120)
        f.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, 3)], seed=21)
        films.append(f.read())
    assert np.array_equal(films[0][0].view(np.uint32), films[1][0].view(np.uint32)) and np.array_equal(films[0][1], films[1][1])
    for name, (o, d) in all_ray_sets(sc.tris, 20000).items():
        rng = np.random.default_rng(17)
        dist = np.where(rng.random(len(o)) < 0.3, rng.random(len(o)), 1.0 + 4.0 * rng.random(len(o))).astype(np.float32)
        want = sc.oracle.shadow_visible(o, d, dist)
        assert np.array_equal(sc.gpu.raycast_shadow(o, d, dist), want), name
        assert np.array_equal(plain.raycast_shadow(o, d, dist), want), name
        idx, t, _, _ = sc.oracle.raycast(o, d)
        bound = np.where(idx >= 0, t * np.float32(1.5), np.float32(1.0)).astype(np.float32)
        bound[::3] = t[::3] * np.float32(0.5)
        want = sc.oracle.shadow_visible(o, d, bound, bounded=True)
        assert np.array_equal(sc.gpu.raycast_shadow(o, d, bound, bounded=True), want), name + " bounded"
        assert np.array_equal(plain.raycast_shadow(o, d, bound, bounded=True), want), name + " bounded"


def test_deferred_dense_triangle_tests(scene_small, scene_medium, monkeypatch):
    """PT_DENSE_SHADOW=1 answers the light-visibility test with k_shadow_dense: leaves are queued in
    LDS instead of tested in place, and dense passes test the queued triangles four lanes per item,
    whatever lane the ray lives in (an option: fewer instructions, but slower on MI355X).  Same
    visibility bit for bit on every ray set, and the same film."""
    monkeypatch.setenv("PT_DENSE_SHADOW", "1")
    dense = {id(sc): pta.GpuScene(sc.host, 0) for sc in (scene_small, scene_medium)}
    monkeypatch.delenv("PT_DENSE_SHADOW")
    for sc in (scene_small, scene_medium):
        g = dense[id(sc)]
        for name, (o, d) in all_ray_sets(sc.tris, 20000).items():
            rng = np.random.default_rng(23)
            dist = np.where(rng.random(len(o)) < 0.3, rng.random(len(o)), 1.0 + 4.0 * rng.random(len(o))).astype(np.float32)
            assert np.array_equal(g.raycast_shadow(o, d, dist), sc.oracle.shadow_visible(o, d, dist)), name
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    films = []
    for g in (scene_medium.gpu, dense[id(scene_medium)]):
        f = pta.Film(g, 320, 200)
        st = f.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, 4)], seed=31)
        films.append((f.read(), (st.closest_rays, st.shadow_rays, st.shadow_rays_traced)))
    assert films[0][1] == films[1][1]
    assert np.array_equal(films[0][0][0].view(np.uint32), films[1][0][0].view(np.uint32)) and np.array_equal(films[0][0][1], films[1][0][1])


@pytest.mark.parametrize("qnodes", [False, True])
def test_wave_wide_leaf_tests(scene_small, scene_medium, monkeypatch, qnodes):
    """PT_LEAFWISE=1 traces with k_trace<MODE, 0>: a lane that reaches a leaf leaves it to the wave,
    which tests it one triangle per lane from a component-by-component copy of the triangles.  Same
    closest hits and the same visibility, bit for bit, on every ray set, in both shadow modes, on
    ordinary leaves and on leaves of up to 200 triangles (more than one 64-triangle piece), and the
    same film.  qnodes: the any-hit walks of these kernels on quantised interior nodes (PT_QNODES=1)."""
    rng = np.random.default_rng(41)
    # 200 copies of a few triangles make leaves that the "no improvement" rule cannot split
    stacked = np.ascontiguousarray(np.concatenate([np.repeat(scene_small.tris[:3], 200 // 3, axis=0), scene_small.tris]))
    stacked_host = pta.HostScene(stacked)
    assert stacked_host.tree_stats()["largest_leaf"] > 64
    stacked_oracle = orc.Scene(stacked)
    monkeypatch.setenv("PT_LEAFWISE", "1")
    if qnodes:
        monkeypatch.setenv("PT_QNODES", "1")
    cases = [(pta.GpuScene(sc.host, 0), sc.oracle, sc.tris) for sc in (scene_small, scene_medium)] + [(pta.GpuScene(stacked_host, 0), stacked_oracle, stacked)]
    monkeypatch.delenv("PT_LEAFWISE")
    monkeypatch.delenv("PT_QNODES", raising=False)
    assert all(g.trace_variant()["qnodes"] == qnodes for g, _, _ in cases[:2])
    for g, oracle, tris in cases:
        for name, (o, d) in all_ray_sets(tris, 20000).items():
            want = oracle.raycast(o, d)
            assert_hits_equal(g.raycast_closest(o, d), want, name + " closest")
            dist = np.where(rng.random(len(o)) < 0.3, rng.random(len(o)), 1.0 + 4.0 * rng.random(len(o))).astype(np.float32)
            assert np.array_equal(g.raycast_shadow(o, d, dist), oracle.shadow_visible(o, d, dist)), name
            idx, t = want[0], want[1]
            bound = np.where(idx >= 0, t * np.float32(1.5), np.float32(1.0)).astype(np.float32)
            bound[::3] = t[::3] * np.float32(0.5)
            assert np.array_equal(g.raycast_shadow(o, d, bound, bounded=True), oracle.shadow_visible(o, d, bound, bounded=True)), name + " bounded"
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    films = []
    for g in (scene_medium.gpu, cases[1][0]):
        f = pta.Film(g, 320, 200)
        st = f.render(cam, pta.cli_lights(), [(0, 0, -1, -1, 0, 4)], seed=31)
        films.append((f.read(), (st.closest_rays, st.shadow_rays, st.shadow_rays_traced)))
    assert films[0][1] == films[1][1]
    assert np.array_equal(films[0][0][0].view(np.uint32), films[1][0][0].view(np.uint32)) and np.array_equal(films[0][0][1], films[1][0][1])
