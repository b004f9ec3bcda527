"""Render rate against samples per render call and the path-numbering unit, on one scene built once.

Usage (GPU box, repo root): python3 scripts/batch_sweep.py [--level 10 --width 3840 --height 2160 --total 64]
Prints one JSON line per (samples per call, PT_PATH_UNIT) pair: traced Mrays/s over `total` samples."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=10)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--total", type=int, default=64)
    ap.add_argument("--spp", default="4,8,16,32,64")
    ap.add_argument("--units", default="4,8")
    args = ap.parse_args()
    import path_tracing_amd as pta
    tris = pta.meshgen.standin_scene(args.level)
    host = pta.HostScene(tris, threads=16)
    gpu = pta.GpuScene(host, 0)
    film = pta.Film(gpu, args.width, args.height)
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5)
    lights = pta.cli_lights()
    film.render(cam, lights, [(0, 0, -1, -1, 0, 4)], seed=1)   # warm-up
    gpu.sync()
    for unit in [int(u) for u in args.units.split(",")]:
        os.environ["PT_PATH_UNIT"] = str(unit)
        for spp in [int(s) for s in args.spp.split(",")]:
            film.zero()
            gpu.sync()
            t0 = time.perf_counter()
            rays = 0
            done = 0
            while done < args.total:
                take = min(spp, args.total - done)
                st = film.render(cam, lights, [(0, 0, -1, -1, done, take)], seed=1, want_stats=True)
                rays += st.closest_rays + st.shadow_rays_traced
                done += take
            gpu.sync()
            dt = time.perf_counter() - t0
            print(json.dumps({"triangles": len(tris), "size": [args.width, args.height], "spp_per_call": spp, "path_unit": unit,
                              "total_spp": args.total, "seconds": round(dt, 4), "ms_per_spp": round(1e3 * dt / args.total, 3),
                              "traced_Mrays_per_s": round(rays / dt / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
