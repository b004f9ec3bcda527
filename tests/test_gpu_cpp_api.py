"""The C++ scene/render API (include/pt_render.h) used directly, as the reference's front ends
other than cli_render use it: single-ray kdTree::ray_test with the hit triangle pointer,
Integrator::get_ray_for_pixel, Integrator::perform_pass on rectangles, Canvas accessors.
tests/native/api_check.cpp prints what it gets; everything is compared with the CPU oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, bits, orc, pta

pytestmark = pytest.mark.gpu


def _f(hexword):
    return struct.unpack("<f", struct.pack("<I", int(hexword, 16)))[0]


def test_cpp_api_against_oracle(tmp_path, scene_small):
    exe = os.path.join(ROOT, "tests", "native", "_build", "api_check")
    assert os.path.exists(exe), "build it with `make native-tests`"
    stl = str(tmp_path / "standin.stl")
    pta.meshgen.write_stl(stl, scene_small.tris)
    W, H = 64, 48
    res = subprocess.run([exe, stl, str(W), str(H)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = res.stdout.splitlines()
    stats = [l for l in lines if l.startswith("stats ")][0].split()
    assert int(stats[3]) == scene_small.oracle.n
    st = scene_small.host.tree_stats()
    assert int(stats[1]) == st["max_depth"] and int(stats[2]) == st["largest_leaf"]

    # single rays: pixel-centre camera rays and their closest hits, bit for bit
    rays = [l.split() for l in lines if l.startswith("ray ")]
    assert len(rays) == len(range(0, H, 7)) * len(range(0, W, 5))
    cam = orc.cli_camera()
    co, cd = orc.camera_rays(cam, W, H)
    co, cd = co.reshape(H, W, 3), cd.reshape(H, W, 3)
    o = np.array([[_f(w) for w in r[3:6]] for r in rays], np.float32)
    d = np.array([[_f(w) for w in r[6:9]] for r in rays], np.float32)
    xy = np.array([[int(r[1]), int(r[2])] for r in rays])
    assert np.array_equal(bits(o), bits(co[xy[:, 1], xy[:, 0]])) and np.array_equal(bits(d), bits(cd[xy[:, 1], xy[:, 0]]))
    want = scene_small.oracle.raycast(o, d)
    got_idx = np.array([int(r[10]) for r in rays], np.int32)
    got_tuv = np.array([[_f(w) for w in r[11:14]] for r in rays], np.float32)
    assert np.array_equal(got_idx, want[0])
    for k in range(3):
        assert np.array_equal(bits(np.ascontiguousarray(got_tuv[:, k])), bits(want[k + 1]))
    assert (got_idx >= 0).any() and (got_idx < 0).any()

    # Integrator::cast_ray: one-path batches, stream (seed 9, call number | 2^31)
    casts = [l.split() for l in lines if l.startswith("cast ")]
    assert len(casts) == 6
    for c in casts:
        k = int(c[1])
        o1 = np.array([[_f(w) for w in c[2:5]]], np.float32)
        d1 = np.array([[_f(w) for w in c[5:8]]], np.float32)
        got = np.array([_f(w) for w in c[9:12]], np.float32)
        want, _ = scene_small.oracle.cast_rays(cam, orc.cli_lights(), o1, d1, ids=[0x80000000 | k], sample=0,
                                               recursions=2 if k == 5 else 10, seed=9)
        assert (np.abs(got - want[0]) <= 1e-5 * np.maximum(np.abs(want[0]), 1.0)).all(), (k, got, want)
    assert any(float(_f(c[9])) > 0 for c in casts)

    # passes: Integrator numbers its passes 0, 1, 2 whatever their rectangle
    assert [l for l in lines if l.startswith("passes ")][0] == "passes 3"
    rgb = np.zeros((H, W, 3), np.float32)
    passes = np.zeros((H, W), np.int32)
    lights = orc.cli_lights()
    for sample, rect in ((0, (8, 4, 24, 16)), (1, (8, 4, 24, 16)), (2, None)):
        scene_small.oracle.render(cam, lights, W, H, rect, sample, 1, seed=5, rgb=rgb, passes=passes)
    px = [l.split() for l in lines if l.startswith("px ")]
    assert len(px) == W * H
    got_p = np.array([int(p[3]) for p in px], np.int32).reshape(H, W)
    got_rgb = np.array([[_f(w) for w in p[4:7]] for p in px], np.float32).reshape(H, W, 3)
    got_8 = np.array([[int(w) for w in p[7:10]] for p in px], np.uint8).reshape(H, W, 3)
    assert np.array_equal(got_p, passes) and got_p.max() == 3 and got_p.min() == 1
    assert (np.abs(got_rgb - rgb) <= 1e-5 * np.maximum(np.abs(rgb), 1.0)).all()
    # Canvas::get_pixel (canvas.cpp:39-45) on the canvas' own sums
    assert np.array_equal(got_8, orc.tonemap(got_rgb, got_p).reshape(H, W, 3))

    # RenderEngine with overlapping rectangles: one engine-wide sample counter from the first
    # overlap on (no pixel sees an index twice), single-ray casts from another thread meanwhile
    conc = [l for l in lines if l.startswith("concurrent ")][0].split()
    assert conc[1] == conc[2] == "200" and conc[3] == "1"
    eng = [l for l in lines if l.startswith("engine ")][0].split()
    assert eng[1:4] == ["7", "0", "1"] and eng[4] == "host"
    rgb = np.zeros((H, W, 3), np.float32)
    passes = np.zeros((H, W), np.int32)
    tw, th = W // 2, H // 2
    order = [None] + [((k & 1) * tw, (k >> 1) * th, tw, th) for k in range(4)] + [None, (0, 0, tw, th)]
    for sample, rect in enumerate(order):
        scene_small.oracle.render(cam, lights, W, H, rect, sample, 1, seed=5, rgb=rgb, passes=passes)
    epx = [l.split() for l in lines if l.startswith("epx ")]
    got_p = np.array([int(p[3]) for p in epx], np.int32).reshape(H, W)
    got_rgb = np.array([[_f(w) for w in p[4:7]] for p in epx], np.float32).reshape(H, W, 3)
    assert np.array_equal(got_p, passes) and got_p.max() == 4 and got_p.min() == 3
    assert (np.abs(got_rgb - rgb) <= 1e-5 * np.maximum(np.abs(rgb), 1.0)).all()


def test_failed_render_is_reported(tmp_path, scene_small):
    """A render call that fails on the GPU (here: a tile of which even one row of 8x8-pixel blocks is
    larger than the batch allowance) must not end in a picture and exit code 0."""
    cli = os.path.join(ROOT, "path_tracing_amd", "bin", "cli_render")
    stl = str(tmp_path / "standin.stl")
    out = str(tmp_path / "never.png")
    pta.meshgen.write_stl(stl, scene_small.tris)
    res = subprocess.run([cli, stl, "--output", out, "--width", "1024", "--height", "64", "--samples", "2", "--tile-width", "1024",
                          "--tile-height", "64"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PT_BATCH_PATHS="4096"))
    assert res.returncode == 1, res.stdout + res.stderr
    assert "Render failed: " in res.stdout and "PT_BATCH_PATHS" in res.stdout
    assert "Wrote to" not in res.stdout and not os.path.exists(out)
