"""The multi-GPU path with the HIP renderer itself: two `gloo` ranks (both on device 0, the test
box has one GPU), each rendering its partition of the sample range into its own film on the GPU;
the reduced film must be the single-rank HIP film (pass counts exactly, energy up to float
summation order) and must agree with the CPU oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, orc, pta

pytestmark = pytest.mark.gpu

W, H, TOTAL, SEED = 200, 136, 7, 11


def _run_ranks(world, out):
    port = 29600 + (os.getpid() % 1000)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        cmd = [sys.executable, os.path.join(ROOT, "tests", "dist_render_worker.py"), "--device", "0", "--width", str(W),
               "--height", str(H), "--total-samples", str(TOTAL), "--seed", str(SEED), "--out", out]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        text, _ = p.communicate(timeout=300)
        assert p.returncode == 0, text


def test_two_hip_ranks_sum_to_the_single_rank_film(tmp_path, scene_small):
    out = str(tmp_path / "reduced.npz")
    _run_ranks(2, out)
    z = np.load(out)
    got = z["film"].reshape(H, W, 4)
    film = pta.Film(scene_small.gpu, W, H)
    tiles = [(x, y, 64, 64, 0, TOTAL) for y in range(0, H, 64) for x in range(0, W, 64)]
    st = film.render(pta.cli_camera(dof_aperture=0.1, dof_distance=4.5), pta.cli_lights(), tiles, seed=SEED)
    rgb, passes = film.read()
    assert np.array_equal(got[..., 3], passes.astype(np.float32)) and (passes == TOTAL).all()
    assert list(z["counts"]) == [st.camera_samples, st.closest_rays, st.shadow_rays]
    assert np.allclose(got[..., :3], rgb, rtol=1e-5, atol=1e-6)
    # and the oracle's picture of the same sample indices
    want, want_p, counts = scene_small.oracle.render(orc.cli_camera(dof_aperture=0.1, dof_distance=4.5), orc.cli_lights(),
                                                     W, H, None, 0, TOTAL, seed=SEED)
    assert np.array_equal(want_p, passes)
    assert (int(counts[0]), int(counts[1])) == (st.closest_rays, st.shadow_rays)
    assert np.allclose(got[..., :3], want, rtol=2e-5, atol=2e-6)
