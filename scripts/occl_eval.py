#!/usr/bin/env python3
"""Counts, on the CPU, what the any-hit walk fetches on the occlusion tree (scripts/occl_build.h) and on
the reference's tree for the shadow rays of a frame sample, and checks that both give the same answers.

    python scripts/occl_eval.py [--level 8] [--width 1920 --height 1080] [--lambertian]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("PT_NO_TORCH", "1")
import orc  # noqa: E402
import path_tracing_amd as pta  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--level", type=int, default=8)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--band-step", type=int, default=64)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--lambertian", action="store_true")
args = ap.parse_args()

tris = pta.meshgen.standin_scene(args.level)
host = pta.HostScene(tris, threads=args.threads)
oscene = orc.Scene(tris, threads=args.threads)
cam = orc.cli_camera(dof_aperture=0.1, dof_distance=4.5, bounce_mode=1 if args.lambertian else 0)
W, H = args.width, args.height
rays = []
for y in range(0, H, args.band_step):
    rays.append(oscene.render_capture(cam, orc.cli_lights(), W, H, (0, y, W, min(8, H - y)), 0, 1, seed=1))
rays8 = np.concatenate(rays)
sh = rays8[(rays8[:, 7] == 1.0)]
o = np.ascontiguousarray(sh[:, 0:3]); d = np.ascontiguousarray(sh[:, 3:6]); dist = np.ascontiguousarray(sh[:, 6])
so = os.path.join(ROOT, "scripts", "_occl_sim.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-I" + os.path.join(ROOT, "include"),
                       "-I" + os.path.join(ROOT, "path_tracing_amd", "csrc"), "-shared", "-fPIC", "-o", so,
                       os.path.join(ROOT, "scripts", "occl_sim.cpp")])
sim = C.CDLL(so)
vis = np.zeros(len(o), np.uint8)
counts = np.zeros(8, np.uint64)
p = lambda a: a.ctypes.data_as(C.c_void_p)
t0 = time.time()
rc = sim.occl_sim(C.byref(host.flat), C.c_uint64(len(o)), p(o), p(d), p(dist), p(vis), p(counts), args.threads)
print("rc", rc, "%.2f s" % (time.time() - t0))
n = max(int(counts[0]), 1)
print("traced shadow rays %d, on the occlusion tree %d, visible %.3f" % (len(o), counts[0], vis.mean()))
print("occlusion tree: depth %d, %d nodes; per ray %.2f node + %.2f triangle records" % (counts[6], counts[7], counts[1] / n, counts[2] / n))
print("reference tree: per ray %.2f node + %.2f triangle records; answers that differ: %d" % (counts[3] / n, counts[4] / n, counts[5]))
want = oscene.shadow_visible(o, d, dist)
print("oracle agrees:", bool(np.array_equal(vis[vis < 2], want[vis < 2])))
