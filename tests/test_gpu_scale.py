"""The large configuration (BASELINE.json configs[4]: about 21 M triangles, 3840x2160) and the
leaf sizes it brings.  The reference caps the tree's depth at 19 (kdtree.cpp:12-15, :67), so at
this size leaves hold 30-50 triangles: the leaf-size field of a child reference overflows (count
0 = "read it from the first record") and a ray walks long leaves.  Everything is checked against
the recursive CPU oracle on its own tree, bit for bit."""
import time

import numpy as np
import pytest

from conftest import assert_hits_equal, bits, orc, oversized_leaf_mesh, pta, rays_over_grid, rays_sphere

pytestmark = pytest.mark.gpu


def test_oversized_leaves_on_the_gpu():
    """A synthetic mesh whose leaves all hold 40 triangles with one common bounding box."""
    tris = oversized_leaf_mesh(24)
    host = pta.HostScene(tris, threads=8)
    oracle = orc.Scene(tris, threads=8)
    st = host.tree_stats()
    assert st == oracle.tree_stats() and st["largest_leaf"] == 40
    gpu = pta.GpuScene(host, 0)
    o, d = rays_over_grid(40000, 24 * 1.5)
    got = gpu.raycast_closest(o, d)
    assert_hits_equal(got, oracle.raycast(o, d), "oversized leaves")
    assert 0.2 < (got[0] >= 0).mean() < 1.0
    rng = np.random.default_rng(9)
    n = len(o)
    dist = np.where(rng.random(n) < 0.5, rng.random(n), 2.0 + rng.random(n)).astype(np.float32)
    assert np.array_equal(gpu.raycast_shadow(o, d, dist), oracle.shadow_visible(o, d, dist))
    assert np.array_equal(gpu.raycast_shadow(o, d, dist, bounded=True), oracle.shadow_visible(o, d, dist, bounded=True))


@pytest.fixture(scope="module")
def huge(request):
    """Level 10: 21 495 808 triangles.  Mesh, host build and oracle build take a couple of minutes
    together; progress goes to the terminal past pytest's capture so the run never looks hung."""
    capman = request.config.pluginmanager.getplugin("capturemanager")

    def say(msg):
        with capman.global_and_fixture_disabled():
            print("[config5] " + msg, flush=True)

    t0 = time.time()
    tris = pta.meshgen.standin_scene(10)
    say("mesh of %d triangles in %.0f s" % (len(tris), time.time() - t0))
    t0 = time.time()
    host = pta.HostScene(tris, threads=16)
    say("host build %.0f s" % (time.time() - t0))
    t0 = time.time()
    oracle = orc.Scene(tris, threads=16)
    say("oracle build %.0f s" % (time.time() - t0))
    return tris, host, oracle, pta.GpuScene(host, 0)


def test_config5_tree_is_the_oracles(huge):
    tris, host, oracle, _ = huge
    assert host.n == 21495808
    st = host.tree_stats()
    assert st == oracle.tree_stats()
    assert st["max_depth"] == 19 and st["largest_leaf"] > 15      # the depth cap at work
    n1, l1 = oracle.tree_dump()
    n2, l2 = host.tree_dump()
    assert np.array_equal(bits(n1), bits(n2)) and np.array_equal(l1, l2)


def test_config5_rays_against_oracle(huge):
    """3840x2160 pixel-centre primaries and incoherent rays: all of them through the GPU, 50 000
    sampled ones of each set against the oracle bit for bit; shadow visibility likewise."""
    tris, host, oracle, gpu = huge
    rng = np.random.default_rng(10)
    W, H = 3840, 2160
    po, pd = orc.camera_rays(orc.cli_camera(), W, H)
    so, sd = rays_sphere(1 << 20, seed=21)
    for name, (o, d) in (("primary 4K", (po, pd)), ("sphere", (so, sd))):
        o = np.ascontiguousarray(o.reshape(-1, 3)); d = np.ascontiguousarray(d.reshape(-1, 3))
        got = gpu.raycast_closest(o, d)
        sel = rng.choice(len(o), 50000, replace=False)
        os_, ds_ = np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel])
        want = oracle.raycast(os_, ds_, threads=16)
        assert_hits_equal([g[sel] for g in got], want, name)
        idx, t, u, v = got
        hit = idx >= 0
        assert 0.3 < hit.mean() < 1.0 and (idx[hit] < host.n).all()
        assert (t[hit] > 0).all() and (u[hit] >= 0).all() and (v[hit] >= 0).all() and (u[hit] + v[hit] <= 1).all()
        # visibility of the same sampled rays: lights nearer than one unit take the exact pass
        dist = np.where(rng.random(len(sel)) < 0.3, rng.random(len(sel)), 1.0 + 4.0 * rng.random(len(sel))).astype(np.float32)
        assert np.array_equal(gpu.raycast_shadow(os_, ds_, dist), oracle.shadow_visible(os_, ds_, dist, threads=16)), name


def test_config5_render_band(huge):
    """Full path trace at 3840x2160 (DOF 0.1/4.5, the three lights): a band of the frame against
    the oracle -- identical ray counts and pass counts, radiance within 1e-5 relative (the bounce
    terms are summed in another order) -- and the band as 64x64 tiles, bit for bit."""
    tris, host, oracle, gpu = huge
    W, H, S = 3840, 2160, 2
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    film = pta.Film(gpu, W, H)
    band = (0, 1072, W, 8)
    st = film.render(cam, pta.cli_lights(), [band + (0, S)], seed=5)
    got, got_p = film.read()
    want, want_p, counts = oracle.render(orc.cli_camera(dof_aperture=0.1, dof_distance=4.5), orc.cli_lights(), W, H, band, 0, S,
                                         seed=5, threads=16)
    assert np.array_equal(got_p, want_p) and got_p.sum() == W * 8 * S
    assert (st.closest_rays, st.shadow_rays) == (int(counts[0]), int(counts[1]))
    err = np.abs(got - want)
    assert (err <= 1e-5 * np.maximum(np.abs(want), 1.0)).all(), err.max()
    film.zero()
    tiles = [(x, 1072, 64, 8, s, 1) for s in range(S) for x in range(0, W, 64)]
    film.render(cam, pta.cli_lights(), tiles, seed=5)
    tiled, _ = film.read()
    assert np.array_equal(tiled.view(np.uint32), got.view(np.uint32))
