#!/usr/bin/env python3
"""What a packet walk (one wave visits a node once for all 64 of its rays) would touch, for camera
rays grouped three ways.  Study tool, CPU only (see packet_model in wave_sim.cpp).

    python scripts/packet_model.py [--level 8] [--packets 2000]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=8)
    ap.add_argument("--packets", type=int, default=2000)
    args = ap.parse_args()
    so = os.path.join(ROOT, "scripts", "_wave_sim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "path_tracing_amd", "csrc"), "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "scripts", "wave_sim.cpp")])
    lib = C.CDLL(so)
    lib.packet_model.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    import path_tracing_amd as pta
    import orc
    host = pta.HostScene(pta.meshgen.standin_scene(args.level), threads=8)
    W, H = 1920, 1080
    cam = orc.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    rng = np.random.default_rng(7)
    n_p = args.packets

    def rays_of(px, py, samples):
        """px, py, samples: equal-length arrays -> jittered DOF camera rays of those (pixel, sample)s."""
        oo = np.empty((len(px), 3), np.float32)
        dd = np.empty((len(px), 3), np.float32)
        for s in np.unique(samples):
            sel = np.nonzero(samples == s)[0]
            # one call per distinct row band keeps this simple: rays of whole rows, then pick
            for y in np.unique(py[sel]):
                row = sel[py[sel] == y]
                o, d = orc.camera_rays(cam, W, H, (0, int(y), W, 1), jittered=True, sample=int(s), seed=1)
                o = o.reshape(-1, 3); d = d.reshape(-1, 3)
                oo[row] = o[px[row]]; dd[row] = d[px[row]]
        return oo, dd

    groupings = {}
    # (a) 64 samples of one pixel
    cx = rng.integers(0, W, n_p); cy = rng.integers(0, H, n_p)
    groupings["64 samples of one pixel"] = (np.repeat(cx, 64), np.repeat(cy, 64), np.tile(np.arange(64), n_p))
    # (b) 4x4 pixels x 4 samples
    bx = rng.integers(0, W // 4, n_p) * 4; by = rng.integers(0, H // 4, n_p) * 4
    k = np.arange(64)
    groupings["4x4 pixels x 4 samples"] = ((bx[:, None] + (k // 4) % 4).ravel(), (by[:, None] + (k // 16)).ravel(), np.tile(k % 4, n_p))
    # (c) 8x8 pixels x 1 sample
    bx = rng.integers(0, W // 8, n_p) * 8; by = rng.integers(0, H // 8, n_p) * 8
    groupings["8x8 pixels x 1 sample"] = ((bx[:, None] + k % 8).ravel(), (by[:, None] + k // 8).ravel(), np.zeros(64 * n_p, np.int64))
    for name, (px, py, ss) in groupings.items():
        o, d = rays_of(np.asarray(px), np.asarray(py), np.asarray(ss))
        o = np.ascontiguousarray(o); d = np.ascontiguousarray(d)
        out = np.zeros(5, np.uint64)
        lib.packet_model(C.byref(host.flat), len(o), o.ctypes.data, d.ctypes.data, out.ctypes.data)
        rays = float(out[4]) * 64
        print("%-26s per ray: %.1f nodes %.1f tris | per packet: %.1f distinct nodes %.1f distinct tris "
              "(x%.2f / x%.2f of one ray's)" % (name, out[0] / rays, out[1] / rays, out[2] / float(out[4]), out[3] / float(out[4]),
                                                 out[2] / float(out[4]) / (out[0] / rays), out[3] / float(out[4]) / max(out[1] / rays, 1e-9)))


if __name__ == "__main__":
    main()
