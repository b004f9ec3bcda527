"""The reference's OWN example front ends -- main.cpp (`example`) and animate.cpp, unmodified -- against
include/pt_render.h: the drop-in claim of the C++ surface, checked on the real callers.

Where the reference's tree is at hand (this container; not the GPU box), `make native-tests` compiles
the two files from where they lie, reading them from standard input so that their `#include
"integrator.h"` / `"visualizer.h"` find two one-line forwarding headers (`#include "pt_render.h"`: the
include swap a maintainer makes), and links them against libpt_render / libpt_host / libpt_hip.  The
CPU test repeats that compilation with -Wall -Werror; the GPU test runs the `example` binary, which
renders 3840x2160 at 1000 samples per pixel into output.png, and holds the picture against
bin/cli_render given the same camera, focus and tile size on the command line."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import ROOT, pta

REF = "/root/reference"
BUILD = os.path.join(ROOT, "tests", "native", "_build")
LIBDIR = os.path.join(ROOT, "path_tracing_amd", "lib")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "main.cpp")), reason="the reference's tree is not on this machine")
@pytest.mark.parametrize("front_end", ["main", "animate"])
def test_reference_front_end_compiles_unmodified(front_end, tmp_path):
    fwd = tmp_path / "fwd"
    fwd.mkdir()
    for header in ("integrator.h", "visualizer.h"):
        (fwd / header).write_text('#include "pt_render.h"\n')
    with open(os.path.join(REF, front_end + ".cpp"), "rb") as src:
        res = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-pthread", "-iquote", str(fwd),
                              "-I" + os.path.join(ROOT, "include"), "-x", "c++", "-", "-o", str(tmp_path / front_end),
                              "-L" + LIBDIR, "-lpt_render", "-lpt_host", "-lpt_hip", "-lz", "-Wl,-rpath," + LIBDIR,
                              "-Wl,-rpath-link,/opt/rocm/lib"], stdin=src, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]


def read_png_rgb(path):
    """RGB8, non-interlaced PNG as written by Canvas::save (filter type 0 on every row)."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", 0
    h = 0
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if kind == b"IHDR":
            w, h, depth, colour = struct.unpack(">IIBB", body[:10])
            assert (depth, colour) == (8, 2)
        elif kind == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert not rows[:, 0].any()
    return rows[:, 1:].reshape(h, w, 3)


@pytest.mark.gpu
def test_reference_example_runs_unmodified(tmp_path, scene_small):
    exe = os.path.join(BUILD, "ref_main")
    if not os.path.exists(exe):
        pytest.skip("tests/native/_build/ref_main was not built (needs the reference's tree at build time)")
    stl = str(tmp_path / "standin.stl")
    pta.meshgen.write_stl(stl, scene_small.tris)
    res = subprocess.run([exe, stl], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    out = res.stdout.replace("\r", "\n")
    assert "Read in %d triangles." % len(scene_small.tris) in out
    assert "Total time to render: " in out and "Triangle tests: 0" in out      # the reference's dead counter stays dead
    rays = [int(l.split(":")[1]) for l in out.splitlines() if l.startswith("Rays cast:")]
    assert rays and rays[0] > 3840 * 2160 * 1000                               # every camera sample casts at least one ray
    got = read_png_rgb(str(tmp_path / "output.png"))
    assert got.shape == (2160, 3840, 3) and got.max() > 0
    # the same picture from cli_render with main.cpp's hard-coded settings spelled out (main.cpp:13-29)
    cli = os.path.join(ROOT, "path_tracing_amd", "bin", "cli_render")
    res = subprocess.run([cli, "--stl", stl, "--output", str(tmp_path / "cli.png"), "--samples", "1000", "--width", "3840", "--height", "2160",
                          "--dof-aperture", "0.1", "--dof-distance", "4.5", "--tile-width", "64", "--tile-height", "64"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    want = read_png_rgb(str(tmp_path / "cli.png"))
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_reference_animate_runs_unmodified(tmp_path, scene_small):
    """animate.cpp as it stands: 100 frames of 1366x768 at 100 samples per pixel into massive/frame%04i.png,
    geometry resident, RenderEngine::zero() and a new focus distance (3.5 + frame / 33.0) per frame.  The
    first and the last frame are the pictures cli_render writes for those settings."""
    exe = os.path.join(BUILD, "ref_animate")
    if not os.path.exists(exe):
        pytest.skip("tests/native/_build/ref_animate was not built (needs the reference's tree at build time)")
    stl = str(tmp_path / "standin.stl")
    pta.meshgen.write_stl(stl, scene_small.tris)
    (tmp_path / "massive").mkdir()
    res = subprocess.run([exe, stl], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    frames = sorted(os.listdir(str(tmp_path / "massive")))
    assert frames == ["frame%04d.png" % k for k in range(100)]
    cli = os.path.join(ROOT, "path_tracing_amd", "bin", "cli_render")
    for frame, focus in ((0, "3.5"), (99, "6.5")):
        res = subprocess.run([cli, "--stl", stl, "--output", str(tmp_path / "cli.png"), "--samples", "100", "--width", "1366", "--height", "768",
                              "--dof-aperture", "0.1", "--dof-distance", focus, "--tile-width", "64", "--tile-height", "64"],
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        got = read_png_rgb(str(tmp_path / "massive" / ("frame%04d.png" % frame)))
        assert got.shape == (768, 1366, 3)
        assert np.array_equal(got, read_png_rgb(str(tmp_path / "cli.png"))), frame
