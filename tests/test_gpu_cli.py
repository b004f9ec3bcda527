"""The drop-in front end: cli_render (reference flags) through the C++ Scene/RenderEngine API
on the GPU, checked against the CPU oracle's image.  BASELINE.json configs[0] geometry:
~1k-triangle STL, 320x240, 16 spp."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, orc, pta

pytestmark = pytest.mark.gpu

CLI = os.path.join(ROOT, "path_tracing_amd", "bin", "cli_render")


def _read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def test_cli_render_matches_oracle(tmp_path, scene_small):
    stl = str(tmp_path / "standin.stl")
    out = str(tmp_path / "out.png")
    pta.meshgen.write_stl(stl, scene_small.tris)
    W, H, S = 320, 240, 16
    cmd = [CLI, stl, "--output", out, "--width", str(W), "--height", str(H), "--samples", str(S),
           "--dof-aperture", "0.1", "--dof-distance", "4.5", "--seed", "4"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "Read in %d triangles." % scene_small.oracle.n in res.stdout
    assert "Wrote to: " + out in res.stdout
    assert "input        = " + stl in res.stdout and "samples      = 16" in res.stdout
    img = _read_png(out)
    assert img.shape == (H, W, 3)
    # the oracle renders the same sample indices: tiles are 64x64, every tile gets samples 0..15
    cam = orc.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    rgb, passes, _ = scene_small.oracle.render(cam, orc.cli_lights(), W, H, None, 0, S, seed=4)
    assert (passes == S).all()
    want = orc.tonemap(rgb, passes).reshape(H, W, 3)
    diff = np.abs(img.astype(int) - want.astype(int))
    # sums differ in the last float bits (bounce terms are added in another order), which can
    # move a value across an 8-bit truncation boundary
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, (diff.max(), (diff > 0).mean())


def test_cli_errors(tmp_path):
    res = subprocess.run([CLI], capture_output=True, text=True, timeout=60)
    assert res.returncode == 1 and "No input STL specified." in res.stdout
    res = subprocess.run([CLI, str(tmp_path / "missing.stl")], capture_output=True, text=True, timeout=60)
    assert res.returncode == 1 and "Couldn't read input file." in res.stdout
    res = subprocess.run([CLI, "--help"], capture_output=True, text=True, timeout=60)
    assert res.returncode == 1 and "--camera-altitude" in res.stdout and "--tile-width" in res.stdout


ANIMATE = os.path.join(ROOT, "path_tracing_amd", "bin", "animate")


def test_animate_frames_match_single_renders(tmp_path, scene_small):
    """The multi-frame driver (reference animate.cpp:26-48): geometry stays resident, the camera
    block and the plane of focus change per frame, every frame restarts its sample stream.  Each
    frame must equal what cli_render produces for that frame's camera on its own."""
    stl = str(tmp_path / "standin.stl")
    pta.meshgen.write_stl(stl, scene_small.tris)
    W, H, S, F = 160, 120, 8, 3
    outdir = str(tmp_path / "frames")
    cmd = [ANIMATE, stl, "--frames", str(F), "--samples", str(S), "--width", str(W), "--height", str(H),
           "--angle-step", "0.3", "--focus-start", "4.0", "--focus-step", "0.25", "--output-dir", outdir, "--seed", "7"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count("Read in %d triangles." % scene_small.oracle.n) == 1   # scene prepared once
    assert "Rays cast: " in res.stdout
    frames = [_read_png(os.path.join(outdir, "frame%04d.png" % f)) for f in range(F)]
    assert not np.array_equal(frames[0], frames[1]) and not np.array_equal(frames[1], frames[2])
    for f in range(F):
        single = str(tmp_path / ("single%d.png" % f))
        cmd = [CLI, stl, "--output", single, "--width", str(W), "--height", str(H), "--samples", str(S),
               "--angle", repr(1.0 + 0.3 * f), "--dof-aperture", "0.1", "--dof-distance", repr(4.0 + 0.25 * f), "--seed", "7"]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stdout + res.stderr
        assert np.array_equal(frames[f], _read_png(single)), f
    # and against the CPU oracle: the last frame's camera (angle 1.6) and plane of focus (4.5) are
    # those the driver set for THAT frame -- the workers re-read them per batch while the geometry
    # stays on the GPU -- and every frame restarts at sample 0
    f = F - 1
    cam = orc.cli_camera(angle=1.0 + 0.3 * f, dof_aperture=0.1, dof_distance=4.0 + 0.25 * f)
    rgb, passes, _ = scene_small.oracle.render(cam, orc.cli_lights(), W, H, None, 0, S, seed=7)
    assert (passes == S).all()
    want = orc.tonemap(rgb, passes).reshape(H, W, 3)
    diff = np.abs(frames[f].astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, (diff.max(), (diff > 0).mean())
    # a frame of another camera is a different picture altogether
    assert np.abs(frames[0].astype(int) - want.astype(int)).max() > 16


def test_checkpoint_and_resume(tmp_path, scene_small):
    """Accumulator dump and resume: 6 samples, stopped after 4 and continued, give the picture
    of 6 samples in one go (pass counts equal; energy sums differ only by float summation order)."""
    stl = str(tmp_path / "standin.stl")
    pta.meshgen.write_stl(stl, scene_small.tris)
    base = [CLI, stl, "--width", "160", "--height", "120", "--dof-aperture", "0.1", "--dof-distance", "4.5", "--seed", "3"]
    whole, part, cont = (str(tmp_path / n) for n in ("whole.png", "part.png", "cont.png"))
    acc4, acc6, acc6b = (str(tmp_path / n) for n in ("acc4.bin", "acc6.bin", "acc6b.bin"))
    for cmd in (base + ["--output", whole, "--samples", "6", "--checkpoint", acc6],
                base + ["--output", part, "--samples", "4", "--checkpoint", acc4],
                base + ["--output", cont, "--samples", "2", "--resume", acc4, "--checkpoint", acc6b]):
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stdout + res.stderr
    assert "Resumed after 4 samples" in res.stdout

    def load(path):
        raw = open(path, "rb").read()
        assert raw[:8] == b"PTACC01\0"
        w, h = np.frombuffer(raw, "<i4", 2, 8)
        rgb = np.frombuffer(raw, "<f4", w * h * 3, 16).reshape(h, w, 3)
        passes = np.frombuffer(raw, "<i4", w * h, 16 + w * h * 12).reshape(h, w)
        return rgb, passes
    rgb_a, p_a = load(acc6)
    rgb_b, p_b = load(acc6b)
    assert (p_a == 6).all() and (p_b == 6).all()
    assert np.allclose(rgb_a, rgb_b, rtol=1e-5, atol=1e-5)
    diff = np.abs(_read_png(whole).astype(int) - _read_png(cont).astype(int))
    assert diff.max() <= 1
    assert not np.array_equal(_read_png(part), _read_png(cont))
    # a checkpoint of another size is refused
    res = subprocess.run(base[:2] + ["--width", "80", "--height", "60", "--samples", "1", "--resume", acc4,
                                     "--output", str(tmp_path / "x.png")], capture_output=True, text=True, timeout=300)
    assert res.returncode == 1 and "Couldn't resume from" in res.stdout


def test_device_side_film_reduce(tmp_path, scene_small):
    """RenderEngine::rebuild_master_canvas through pt_films_reduce: with PT_FILMS_REDUCE=rccl the
    films are summed by an RCCL reduce on the device side even on one GPU (a one-rank
    communicator); the picture must be the one the plain read-back gives."""
    stl = str(tmp_path / "standin.stl")
    pta.meshgen.write_stl(stl, scene_small.tris)
    outs = []
    for how in ("host", "rccl"):
        out = str(tmp_path / (how + ".png"))
        cmd = [CLI, stl, "--output", out, "--width", "200", "--height", "120", "--samples", "3", "--seed", "2"]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, PT_FILMS_REDUCE=how))
        assert res.returncode == 0, res.stdout + res.stderr
        assert "reduce failed" not in res.stderr, res.stderr
        assert "films summed by: " + how in res.stderr, res.stderr       # the collective itself, not a fallback
        outs.append(_read_png(out))
    assert np.array_equal(outs[0], outs[1])


def test_two_workers_share_the_queue(tmp_path, scene_small):
    """The several-worker machinery on a one-GPU box (PT_ENGINE_WORKERS=2: two workers, each with
    its own film, both on GPU 0): runs of tile passes are split between the workers at batch
    boundaries, every pass is rendered exactly once, and the summed films give the picture of one
    worker (sums associate differently: at most one 8-bit level) and of the oracle."""
    stl = str(tmp_path / "standin.stl")
    pta.meshgen.write_stl(stl, scene_small.tris)
    W, H, S = 320, 200, 24
    outs = {}
    for workers, batch in (("1", None), ("2", "65536"), ("3", "40000")):
        out = str(tmp_path / ("w%s.png" % workers))
        acc = str(tmp_path / ("w%s.acc" % workers))
        env = dict(os.environ, PT_ENGINE_WORKERS=workers)
        if batch:
            env["PT_ENGINE_BATCH_PATHS"] = batch      # small batches: many hand-overs between the workers
        cmd = [CLI, stl, "--output", out, "--width", str(W), "--height", str(H), "--samples", str(S), "--dof-aperture", "0.1",
               "--dof-distance", "4.5", "--seed", "6", "--checkpoint", acc]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        assert res.returncode == 0, res.stdout + res.stderr
        raw = open(acc, "rb").read()
        passes = np.frombuffer(raw, "<i4", W * H, 16 + W * H * 12)
        assert (passes == S).all(), (workers, passes.min(), passes.max())
        outs[workers] = _read_png(out)
    for workers in ("2", "3"):
        diff = np.abs(outs["1"].astype(int) - outs[workers].astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, (workers, diff.max(), (diff > 0).mean())
    rgb, passes, _ = scene_small.oracle.render(orc.cli_camera(dof_aperture=0.1, dof_distance=4.5), orc.cli_lights(), W, H, None, 0, S, seed=6)
    want = orc.tonemap(rgb, passes).reshape(H, W, 3)
    diff = np.abs(outs["2"].astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3


def test_forced_collective_with_shared_gpu_is_refused(tmp_path, scene_small):
    """Two workers on ONE GPU and the RCCL reduce asked for by name: the collective sums one film per
    GPU, so rebuild_master_canvas must refuse cleanly (non-zero exit, reason on stderr) -- not hang
    on the GPU's lock, not substitute the host sum."""
    stl = str(tmp_path / "standin.stl")
    pta.meshgen.write_stl(stl, scene_small.tris)
    out = str(tmp_path / "refused.png")
    cmd = [CLI, stl, "--output", out, "--width", "160", "--height", "96", "--samples", "2", "--seed", "2"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, PT_ENGINE_WORKERS="2", PT_FILMS_REDUCE="rccl"))
    assert res.returncode != 0, res.stdout + res.stderr
    assert "one film per GPU required" in res.stderr, res.stderr
