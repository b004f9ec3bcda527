#!/usr/bin/env python3
"""Compares wave scheduling policies of the trace kernel on the CPU (see wave_sim.cpp).

    python scripts/wave_sim.py [--level 7] [--rays 40000]
Prints, per ray set and policy, the modelled instruction cost per ray and the lane utilisation.
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


class SimParams(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("policy", "refill_min", "tri_threshold", "max_pops", "tris_per_round", "any_hit")] + \
               [(k, C.c_double) for k in ("c_round", "c_tri", "c_pop", "c_node", "c_start")]


class SimStats(C.Structure):
    _fields_ = [("cost", C.c_double)] + [(k, C.c_uint64) for k in (
        "rounds", "tri_sections", "pop_sections", "node_sections", "start_sections",
        "lane_tri", "lane_pop", "lane_node", "lane_pop_culled")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=7)
    ap.add_argument("--rays", type=int, default=40000)
    ap.add_argument("--costs", default="now", help="now | lean")
    args = ap.parse_args()
    so = os.path.join(ROOT, "scripts", "_wave_sim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "path_tracing_amd", "csrc"), "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "scripts", "wave_sim.cpp")])
    lib = C.CDLL(so)
    lib.wave_sim.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(SimParams), C.POINTER(SimStats)]
    lib.wave_sim_two.argtypes = lib.wave_sim.argtypes
    import path_tracing_amd as pta
    import orc
    host = pta.HostScene(pta.meshgen.standin_scene(args.level), threads=8)
    W, H = 1920, 1080
    rows = max(1, args.rays // W)
    o, d = orc.camera_rays(orc.cli_camera(), W, H, (0, 500, W, rows))
    o = np.ascontiguousarray(o.reshape(-1, 3), np.float32)
    d = np.ascontiguousarray(d.reshape(-1, 3), np.float32)
    rng = np.random.default_rng(1234)
    m = args.rays
    so_ = rng.normal(size=(m, 3)); so_ = 3 * so_ / np.linalg.norm(so_, axis=1, keepdims=True)
    tt = rng.normal(size=(m, 3)); tt = tt / np.linalg.norm(tt, axis=1, keepdims=True) * rng.random((m, 1)) ** (1 / 3)
    sd = (tt - so_).astype(np.float32); sd /= np.linalg.norm(sd, axis=1, keepdims=True)
    sets = {"primary": (o, d), "sphere": (np.ascontiguousarray(so_, np.float32), np.ascontiguousarray(sd, np.float32))}
    costs = {"now": dict(c_round=40, c_tri=80, c_pop=45, c_node=215, c_start=70),
             "lean": dict(c_round=40, c_tri=72, c_pop=20, c_node=97, c_start=77)}[args.costs]
    policies = [("kernel: T|N then pops", dict(policy=4, max_pops=2, tris_per_round=1, tri_threshold=0))]
    for thr in (16, 32, 40, 48, 56):
        policies.append(("either thr=%d" % thr, dict(policy=1, max_pops=2, tris_per_round=1, tri_threshold=thr)))
    for sel in (0, 10, 20):
        for pops in (1, 2):
            policies.append(("two rays/lane sel=%d pops=%d" % (sel, pops), dict(policy=5, max_pops=pops, tris_per_round=1, tri_threshold=sel)))
    policies[0] = ("kernel: T|N then 1 pop", dict(policy=4, max_pops=1, tris_per_round=1, tri_threshold=0))
    for name, (oo, dd) in sets.items():
        n = len(oo)
        print("== %s: %d rays" % (name, n))
        for pname, kw in policies:
            p = SimParams(refill_min=8, any_hit=0, **kw, **costs)
            s = SimStats()
            refill = 16 if kw["policy"] != 5 else 32
            p.refill_min = refill
            (lib.wave_sim_two if kw["policy"] == 5 else lib.wave_sim)(C.byref(host.flat), n, oo.ctypes.data, dd.ctypes.data, C.byref(p), C.byref(s))
            useful = s.lane_tri * p.c_tri + s.lane_pop * p.c_pop + s.lane_node * p.c_node
            print("  %-28s cost/ray %7.1f  rounds/ray*64 %6.1f  T/P/N sections per ray*64 %5.1f %5.1f %5.1f  "
                  "lane steps/ray T %.1f P %.1f (culled %.1f) N %.1f  util %.2f" % (
                      pname, s.cost / n, s.rounds / n * 64, s.tri_sections / n * 64, s.pop_sections / n * 64,
                      s.node_sections / n * 64, s.lane_tri / n, s.lane_pop / n, s.lane_pop_culled / n, s.lane_node / n,
                      useful / 64 / s.cost))


if __name__ == "__main__":
    main()
