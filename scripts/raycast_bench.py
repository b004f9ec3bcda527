#!/usr/bin/env python3
"""Micro-benchmark of the trace kernels alone (BASELINE.json configs[1] and an incoherent set).

Rays are resident in HBM; time is the kernels' own (HIP events on the library's stream).
    python scripts/raycast_bench.py [--level 8] [--reps 5] [--sets primary,bounce,sphere,shadow,binned]
"binned" adds, for every incoherent set, the same rays ordered by direction octant and Morton cell
of the origin (and shuffled), to price what a ray-binning pass could buy.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sets", default="primary,sphere,shadow")
    ap.add_argument("--sphere-rays", type=int, default=1 << 21)
    ap.add_argument("--check", action="store_true", help="compare a sample of results with the CPU oracle")
    ap.add_argument("--check-emul", type=int, default=0, help="compare this many primary rays with the CPU lane emulation")
    args = ap.parse_args()

    import torch
    import path_tracing_amd as pta

    tris = pta.meshgen.standin_scene(args.level)
    import time
    t0 = time.time()
    host = pta.HostScene(tris, threads=16)
    t_host = time.time() - t0
    gpu = pta.GpuScene(host, 0)
    W, H = args.width, args.height
    n = W * H
    dev = "cuda"
    res = {"triangles": host.n, "tree": host.tree_stats(), "host_prepare_s": round(t_host, 2), "host_timings": host.timings()}

    def soa(a):
        return [torch.from_numpy(np.ascontiguousarray(a[:, k])).to(dev) for k in range(3)]

    def run_closest(name, rays):
        m = rays[0].numel()
        idx = torch.empty(m, dtype=torch.int32, device=dev)
        t = torch.empty(m, dtype=torch.float32, device=dev)
        u = torch.empty_like(t)
        v = torch.empty_like(t)
        torch.cuda.synchronize()
        gpu.raycast_closest_dev(rays, idx, t, u, v)
        gpu.sync()
        gpu.kernel_time(0, reset=True)
        for _ in range(args.reps):
            gpu.raycast_closest_dev(rays, idx, t, u, v)
        gpu.sync()
        ms, launches = gpu.kernel_time(0, reset=True)
        res[name] = {"rays": m, "ms_per_launch": round(ms / launches, 4), "Mrays_s": round(m / (ms / launches) / 1e3, 1),
                     "hit_fraction": round(float((idx >= 0).float().mean().item()), 4)}
        return idx, t, u, v

    sets = args.sets.split(",")
    cam = pta.cli_camera()
    prim = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(6)]
    gpu.camera_rays_dev(cam, W, H, (0, 0, W, H), 0, 0, 1, prim)
    gpu.sync()
    if "primary" in sets:
        out = run_closest("primary_rowmajor", prim)
        if args.check:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import orc
            osc = orc.Scene(tris, threads=16)
            sel = np.random.default_rng(0).choice(n, 200000, replace=False)
            o = np.stack([p.cpu().numpy()[sel] for p in prim[:3]], axis=1)
            d = np.stack([p.cpu().numpy()[sel] for p in prim[3:]], axis=1)
            want = osc.raycast(o, d, threads=16)
            got = [x.cpu().numpy()[sel] for x in out]
            res["primary_check_bit_exact"] = bool(all(
                np.array_equal(np.ascontiguousarray(g).view(np.uint32), np.ascontiguousarray(w).view(np.uint32))
                for g, w in zip(got, want)))
        if args.check_emul:
            # Scenes too large for the CPU oracle: compare with the lane-level CPU emulation of
            # the same device logic over the same flat arrays (tests/native/traverse_emul.cpp),
            # which the CPU test-suite holds bit-exact against the oracle on smaller scenes.
            import ctypes as C
            E = C.CDLL(os.path.join(ROOT, "tests", "native", "_build", "libtraverse_emul.so"))
            E.emul_raycast_closest.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 7
            sel = np.random.default_rng(0).choice(n, args.check_emul, replace=False)
            o = np.ascontiguousarray(np.stack([p.cpu().numpy()[sel] for p in prim[:3]], axis=1))
            d = np.ascontiguousarray(np.stack([p.cpu().numpy()[sel] for p in prim[3:]], axis=1))
            m = len(sel)
            w_idx = np.empty(m, np.int32)
            w_t, w_u, w_v = (np.empty(m, np.float32) for _ in range(3))
            flat = host.flat
            ptr = lambda a: a.ctypes.data_as(C.c_void_p)
            worst = E.emul_raycast_closest(C.byref(flat), m, ptr(o), ptr(d), ptr(w_idx), ptr(w_t), ptr(w_u), ptr(w_v), None)
            got = [x.cpu().numpy()[sel] for x in out]
            res["primary_check_vs_emulation_bit_exact"] = bool(all(
                np.array_equal(np.ascontiguousarray(g).view(np.uint32), np.ascontiguousarray(w).view(np.uint32))
                for g, w in zip(got, (w_idx, w_t, w_u, w_v))))
            res["emulation_max_stack_dwords"] = int(worst)
    if "lockstep" in sets:
        # every ray 64 times in a row: a wave's lanes walk in lockstep, so each round runs ONE of the
        # kernel's phases (node / triangle / pop) with all lanes -- what phase mixing costs otherwise
        k = n // 64
        sel = torch.arange(k, device=dev) * 64
        rep = [p[sel].repeat_interleave(64).contiguous() for p in prim]
        run_closest("primary_each_ray_x64", rep)
        run_closest("primary_subset_once", [p[sel].contiguous() for p in prim])
    if "sphere" in sets:
        rng = np.random.default_rng(1234)
        m = args.sphere_rays
        o = rng.normal(size=(m, 3))
        o = 3 * o / np.linalg.norm(o, axis=1, keepdims=True)
        tt = rng.normal(size=(m, 3))
        tt = tt / np.linalg.norm(tt, axis=1, keepdims=True) * rng.random((m, 1)) ** (1 / 3)
        d = (tt - o).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = soa(o.astype(np.float32)) + soa(d.astype(np.float32))
        run_closest("incoherent_sphere", rays)
    def binned(rays):
        """The same rays ordered by direction octant, then by a 30-bit Morton code of the origin's
        cell in a 1024^3 grid over the rays' bounding box: what a binning pass ahead of the trace
        kernel could deliver at best."""
        o = torch.stack(rays[:3], dim=1)
        d = torch.stack(rays[3:], dim=1)
        lo, hi = o.min(dim=0).values, o.max(dim=0).values
        cell = ((o - lo) / (hi - lo).clamp_min(1e-20) * 1023.0).clamp(0, 1023).to(torch.int64)

        def spread(x):
            x = (x | (x << 16)) & 0x030000FF
            x = (x | (x << 8)) & 0x0300F00F
            x = (x | (x << 4)) & 0x030C30C3
            x = (x | (x << 2)) & 0x09249249
            return x
        morton = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)
        octant = (d[:, 0] < 0).to(torch.int64) | ((d[:, 1] < 0).to(torch.int64) << 1) | ((d[:, 2] < 0).to(torch.int64) << 2)
        order = torch.argsort((octant << 30) | morton)
        return [r[order].contiguous() for r in rays]

    if "sphere" in sets and "binned" in sets:
        run_closest("incoherent_sphere_binned", binned(rays))
    if "bounce" in sets:
        # mirror-bounce rays off the primary hits (geometric normal), in screen order and binned
        idx, t, u, v = run_closest("primary_for_bounce", prim)
        hit = idx >= 0
        nrm = torch.from_numpy(np.ascontiguousarray(host.triangle_data()[:, 15:18])).to(dev)[idx[hit].long()]
        dd = torch.stack([prim[3 + k][hit] for k in range(3)], dim=1)
        oo = torch.stack([prim[k][hit] + t[hit] * prim[3 + k][hit] for k in range(3)], dim=1)
        nd = (nrm * dd).sum(dim=1, keepdim=True)
        nrm = torch.where(nd > 0, -nrm, nrm)
        refl = dd - 2.0 * (nrm * dd).sum(dim=1, keepdim=True) * nrm
        refl = refl / refl.norm(dim=1, keepdim=True)
        oo = oo + 1e-3 * nrm
        brays = [oo[:, k].contiguous() for k in range(3)] + [refl[:, k].contiguous() for k in range(3)]
        run_closest("bounce_screen_order", brays)
        # a second bounce: far less coherent than the first
        idx2, t2, _, _ = run_closest("bounce_screen_order", brays)
        if "binned" in sets:
            run_closest("bounce_binned", binned(brays))
        perm = torch.randperm(brays[0].numel(), device=dev)
        run_closest("bounce_shuffled", [r[perm].contiguous() for r in brays])
    if "shadow" in sets:
        # shadow-like rays: from primary hit points toward the first light
        idx, t, u, v = run_closest("primary_again", prim) if "primary" not in sets else out
        hit = idx >= 0
        o = torch.stack([prim[k][hit] + t[hit] * prim[3 + k][hit] for k in range(3)], dim=1)
        lp = torch.tensor([0.0, 0.0, 3.0], device=dev)
        d = lp[None, :] - o
        dist = d.norm(dim=1)
        d = d / dist[:, None]
        o = o + 1e-3 * d
        rays = [o[:, k].contiguous() for k in range(3)] + [d[:, k].contiguous() for k in range(3)]
        m = rays[0].numel()
        vis = torch.empty(m, dtype=torch.uint8, device=dev)
        distc = dist.contiguous()
        torch.cuda.synchronize()
        gpu.raycast_shadow_dev(rays, distc, vis)
        gpu.sync()
        gpu.kernel_time(1, reset=True)
        for _ in range(args.reps):
            gpu.raycast_shadow_dev(rays, distc, vis)
        gpu.sync()
        ms, launches = gpu.kernel_time(1, reset=True)
        res["shadow_to_light0"] = {"rays": m, "ms_per_call": round(ms / args.reps, 4),
                                   "Mrays_s": round(m / (ms / args.reps) / 1e3, 1),
                                   "visible_fraction": round(float((vis == 1).float().mean().item()), 4)}
        if "binned" in sets:
            perm = torch.randperm(m, device=dev)
            for name, rr, dd2 in (("shadow_to_light0_shuffled", [r[perm].contiguous() for r in rays], distc[perm].contiguous()),):
                gpu.raycast_shadow_dev(rr, dd2, vis)
                gpu.sync()
                gpu.kernel_time(1, reset=True)
                for _ in range(args.reps):
                    gpu.raycast_shadow_dev(rr, dd2, vis)
                gpu.sync()
                ms, launches = gpu.kernel_time(1, reset=True)
                res[name] = {"rays": m, "ms_per_call": round(ms / args.reps, 4), "Mrays_s": round(m / (ms / args.reps) / 1e3, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
