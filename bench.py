#!/usr/bin/env python3
"""Benchmark of the ray-casting hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its N ranks itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric "Mrays/s (all bounces)" is
quoted on): full path trace of the ~1.3 M-triangle stand-in object on its floor, 3 lights,
mirror bounces (the reference's HEAD behaviour), Gaussian-aperture depth of field,
1920x1080.

Default (weak scaling): one step = one batch of --spp-per-step full-frame sample passes per GPU;
every rank renders its own sample range into its own film and the films are summed to rank 0 by
one RCCL reduce inside the timed region.  After the timed steps the same processes render ONE
frame of --strong-spp samples (default 1000) split over the ranks by sample index
(path_tracing_amd.distributed.partition_samples), reduce included, and report its wall time as
`strong_scaling` -- the "1080p @ 1000 spp on N GPUs" figure.  With --total-spp T that frame IS the
step (strong scaling: the work per step is fixed, each rank renders T/N samples).

Prints ONE JSON line on rank 0.  `value` counts the rays that were traced -- closest-hit casts
plus the shadow rays of light samples with a non-zero weight -- per second of wall time, with
scene and film resident in HBM before the clock starts; `reference_counted_Mrays_per_s` counts
what the reference's `rays_cast++` (kdtree.cpp:578) would, i.e. also the shadow rays of
zero-weight light samples, which are counted but not traced.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
# Ceiling for a working set that lives in L2 / Infinity Cache: what the compute units' vector-memory
# path delivers for gathers of L2-resident rows, chip-wide (hardware guide: 16.8-18.8 TB/s measured;
# the L2's aggregate bandwidth is 34.5 TB/s).  This repository's own micro-benchmark of the exact
# access shape (64 lanes each loading their own 64-byte record) gets 13.4 TB/s.
L2_GATHER_PEAK_GBS = 16800.0
L2_AGGREGATE_GBS = 34500.0
OWN_GATHER_RATE_GBS = 13400.0
INFINITY_CACHE_BYTES = 256 << 20


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--level", type=int, default=8, help="icosphere subdivision level of the stand-in mesh (8 -> 1.31 M)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-step", type=int, default=256, help="weak mode: one step = this many full-frame sample passes per GPU; "
                                                                   "also the largest batch handed to one render call")
    ap.add_argument("--total-spp", type=int, default=0, help="strong mode: one step = one frame of this many samples split over the ranks")
    ap.add_argument("--strong-spp", type=int, default=1000, help="weak mode: samples of the extra strong-scaling frame (0: skip it)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--lambertian", action="store_true",
                    help="the flagged Lambertian bounce mode (reference integrator.cpp:79-84 without :104) instead of mirror bounces")
    ap.add_argument("--dist-one-rank", action="store_true",
                    help="with one rank, still set up the process group and go through every collective of the N-rank path "
                         "(RCCL with a single member): exercises the nccl calls on a one-GPU box")
    ap.add_argument("--rehearse", action="store_true",
                    help="run the N-rank code path on ONE GPU: gloo backend, every rank on device 0, films reduced through host copies")
    return ap.parse_args()


def host_threads(world):
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 8
    env = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if env > 0:
        avail = min(avail, env)
    return max(1, min(16, avail // max(world, 1)) or 1)


def measured_traffic(key):
    """HBM-side bytes per ray of the trace kernels for workload `key`, from the rocprofv3 PMC passes
    recorded in profiles/traffic.json (FETCH_SIZE + WRITE_SIZE per ray of the same bench command,
    FETCH_SIZE doubled as the gfx950 correction requires)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fp:
            t = json.load(fp)
        return t.get("configs", {}).get(key)
    except (OSError, ValueError):
        return None


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process starts the N
    ranks itself, one fresh child per GPU (the reference's RenderEngine spawns its N workers
    itself, integrator.cpp:361-378), before anything here has imported torch or touched a GPU.
    It relays rank 0's JSON line and exits non-zero if any rank does."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    line0 = b""
    try:
        line0 = procs[0].stdout.read()      # rank 0 prints the one JSON line; stderr passes through
        pending = list(procs)
        status = 0
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 1
                    # one rank failed: the others would wait for it in the next collective
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    # only the JSON line goes to stdout; whatever else rank 0 or its libraries printed there goes to stderr
    for text in line0.decode(errors="replace").splitlines():
        is_result = text.startswith("{") and status == 0
        print(text, file=sys.stdout if is_result else sys.stderr, flush=True)
    if status != 0:
        print("[bench] a rank exited with status %d" % status, file=sys.stderr)
    return status


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print("[bench] --gpus %d but the launcher started %d rank(s): running %d" % (args.gpus, world, world), file=sys.stderr)
    import torch
    import torch.distributed as dist

    import path_tracing_amd as pta
    from path_tracing_amd.distributed import partition_samples, reduce_film

    device = 0 if args.rehearse else local_rank
    torch.cuda.set_device(device)
    use_dist = world > 1 or args.dist_one_rank
    if use_dist:
        if "MASTER_ADDR" not in os.environ:      # one rank started by hand
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ["MASTER_ADDR"] = "127.0.0.1"
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))

    def note(msg):
        # progress on stderr: big scenes spend minutes before the first kernel
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    W, H, spp = args.width, args.height, args.spp_per_step
    strong = args.total_spp > 0
    t0 = time.time()
    tris = pta.meshgen.standin_scene(args.level)
    note("mesh of %d triangles generated in %.1f s" % (len(tris), time.time() - t0))
    t0 = time.time()
    host = pta.HostScene(tris, threads=host_threads(world))
    t_prepare = time.time() - t0
    note("normals, tree and flat arrays in %.1f s" % t_prepare)
    gpu = pta.GpuScene(host, device)
    film_t = torch.zeros(H * W * 4, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    film = pta.Film(gpu, W, H, external=film_t)
    cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5, bounce_mode=1 if args.lambertian else 0)
    lights = pta.cli_lights()

    def reduce_to_rank0():
        """Canvas::add_from over all workers (canvas.cpp:47-52): one sum-reduce of the film."""
        if not use_dist:
            return
        if args.rehearse:
            h = film_t.cpu()
            reduce_film(h, dst=0)
            if rank == 0:
                film_t.copy_(h)
        else:
            reduce_film(film_t, dst=0)

    class Tally:
        def __init__(self):
            self.closest = self.shadow = self.traced = self.samples = 0

        def add(self, st):
            self.closest += st.closest_rays
            self.shadow += st.shadow_rays
            self.traced += st.shadow_rays_traced
            self.samples += st.camera_samples

    def render_range(first, count, tally):
        done = 0
        while done < count:
            take = min(spp, count - done)
            tally.add(film.render(cam, lights, [(0, 0, -1, -1, first + done, take)], seed=args.seed, want_stats=True))
            done += take

    def frame(total, base, tally):
        """One frame of `total` samples per pixel, split over the ranks by sample index."""
        first, count = partition_samples(total, world, rank)
        # one call with all of the rank's samples: the library cuts it into screen bands that each
        # carry every sample (more samples of a patch in flight = fewer distinct leaves touched)
        if count:
            tally.add(film.render(cam, lights, [(0, 0, -1, -1, base + first, count)], seed=args.seed, want_stats=True))

    total_steps = args.warmup + args.steps

    def step(i, tally):
        if strong:
            frame(args.total_spp, i * args.total_spp, tally)
        else:
            render_range((rank * total_steps + i) * spp, spp, tally)

    def all_sum(values):
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        if use_dist:
            if not args.rehearse:
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(x) for x in t.cpu()]

    def all_max(value):
        t = torch.tensor([float(value)], dtype=torch.float64)
        if use_dist:
            if not args.rehearse:
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.cpu()[0])

    scratch = Tally()
    for i in range(args.warmup):
        step(i, scratch)
    gpu.sync()
    note("warm-up done")
    reduce_to_rank0()  # untimed: first use of the collective (communicator, kernels)
    torch.cuda.synchronize()
    film.zero()
    for c in range(6):
        gpu.kernel_time(c, reset=True)
    gpu.sync()

    # ---- the timed steps ----
    barrier()
    t_start = time.perf_counter()
    tally = Tally()
    for i in range(args.warmup, total_steps):
        step(i, tally)
    gpu.sync()
    reduce_to_rank0()
    barrier()
    elapsed = all_max(time.perf_counter() - t_start)
    closest_all, shadow_all, samples_all, traced_all = all_sum([tally.closest, tally.shadow, tally.samples, tally.traced])
    kt = {c: gpu.kernel_time(c) for c in range(6)}
    film_ok = None
    if rank == 0:
        # sanity of the reduced film: every pixel got every rank's passes
        passes = film_t.view(H, W, 4)[..., 3]
        expect = float(args.steps * (args.total_spp if strong else spp * world))
        film_ok = bool((passes == expect).all().item())

    # ---- weak mode: the strong-scaling frame (1080p @ 1000 spp on N GPUs), timed on its own ----
    strong_extra = None
    if not strong and args.strong_spp > 0:
        film.zero()
        gpu.sync()
        barrier()
        t1 = time.perf_counter()
        st = Tally()
        frame(args.strong_spp, 0, st)
        gpu.sync()
        reduce_to_rank0()
        barrier()
        wall = all_max(time.perf_counter() - t1)
        c_all, s_all, n_all, tr_all = all_sum([st.closest, st.shadow, st.samples, st.traced])
        if rank == 0:
            ok = bool((film_t.view(H, W, 4)[..., 3] == float(args.strong_spp)).all().item())
            strong_extra = {"what": "one %dx%d frame of %d samples per pixel, split over the %d rank(s) by sample index, film reduce included"
                                    % (W, H, args.strong_spp, world),
                            "wall_s": round(wall, 4), "n_gpus": world, "spp_total": args.strong_spp,
                            "spp_per_rank": [partition_samples(args.strong_spp, world, r)[1] for r in range(world)],
                            "camera_samples": int(n_all), "traced_Mrays_per_s": round((c_all + tr_all) / wall / 1e6, 2),
                            "film_check": ok}

    if rank == 0:
        rays_ref = closest_all + shadow_all
        rays_traced = closest_all + traced_all
        value = rays_traced / elapsed / 1e6
        std_cfg = (args.level, W, H) == (8, 1920, 1080) and not args.lambertian
        per_step = ("%d spp per step split over the ranks" % args.total_spp) if strong else ("%d spp per step per GPU" % spp)
        out = {
            "metric": "Mrays/s (all bounces)", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "full path trace (" + ("configs[2]" if std_cfg else "other scene or size") +
                            "): stand-in bumpy icosphere level %d + floor, %d triangles, "
                            "3 lights, %s <= 10, DOF 0.1/4.5, %dx%d, %s"
                            % (args.level, host.n, "LAMBERTIAN bounces (flagged extension: integrator.cpp:79-84 without :104)"
                               if args.lambertian else "mirror bounces", W, H, per_step),
                "triangles": host.n, "width": W, "height": H, "spp_per_step": args.total_spp if strong else spp,
                "parallelism": "sample-partitioned x%d, one %s reduce of the film" % (world, "gloo (rehearsal on one GPU)" if args.rehearse else "RCCL"),
            },
            "value_definition": "rays traced per second of wall time: closest-hit casts + shadow casts of light samples with "
                                "non-zero weight; reference_counted_Mrays_per_s is the other definition (every cast the reference makes)",
            "ray_counting": "value = rays traced per second (closest-hit casts + shadow rays of light samples with non-zero weight); "
                            "reference_counted_Mrays_per_s also counts the shadow rays of zero-weight light samples, which the "
                            "reference casts and this renderer counts but does not trace (same picture bit for bit)",
            "reference_counted_Mrays_per_s": round(rays_ref / elapsed / 1e6, 2),
            "camera_samples_per_s": round(samples_all / elapsed, 1),
            "wall_s_1080p_1000spp_extrapolated": round(elapsed / samples_all * 1920 * 1080 * 1000, 2),
            "rays_per_sample": round(rays_ref / samples_all, 4),
            "closest_rays": int(closest_all), "shadow_rays": int(shadow_all), "camera_samples": int(samples_all),
            "shadow_rays_traced": int(traced_all),
            "film_check": film_ok,
            "strong_scaling": strong_extra,
            "scene_prepare_s": round(t_prepare, 2),
            "kernel_ms": {name: {"ms": round(kt[c][0], 3), "launches": kt[c][1]} for c, name in
                          enumerate(["closest", "shadow", "generate", "shade", "accumulate", "shadow_exact_pass"])},
        }

        # ---- CPU oracle: work per ray in reference visit order, and (N=1) the CPU baseline ----
        roof = None
        cpu = None
        if not args.no_cpu_baseline:
            import ctypes as C
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import orc
            threads = args.cpu_threads or host_threads(1)
            tb = time.time()
            oscene = orc.Scene(tris, threads=threads)
            t_oracle_build = time.time() - tb
            note("oracle scene in %.1f s" % t_oracle_build)
            ocam = orc.cli_camera(dof_aperture=0.1, dof_distance=4.5, bounce_mode=cam.bounce_mode)
            # Work per ray on a sample of the frame (every 8th 8-row band, 1 spp):
            #  - box and triangle tests of the reference's visit order (the oracle's counting walk), and
            #  - the records the device walks fetch: the rays the oracle casts on the sample, replayed
            #    through the kernels' own lane logic (csrc/traverse_core.h compiled for the CPU,
            #    tests/native/traverse_emul.cpp) with a counter on every record load.
            emul = C.CDLL(os.path.join(ROOT, "tests", "native", "_build", "libtraverse_emul.so"))
            emul.emul_work_counts.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
            bands = [(0, y, W, min(8, H - y)) for y in range(0, H, 64)]
            work = np.zeros(7, dtype=np.uint64)
            fetched = np.zeros(12, dtype=np.uint64)
            rgb = np.zeros((H, W, 3), np.float32)
            pas = np.zeros((H, W), np.int32)
            flat = host.flat
            for b in bands:
                _, _, c = oscene.render(ocam, orc.cli_lights(), W, H, b, 0, 1, seed=args.seed, threads=threads,
                                        rgb=rgb, passes=pas, count_work=True)
                work += c
                rays8 = oscene.render_capture(ocam, orc.cli_lights(), W, H, b, 0, 1, seed=args.seed)
                emul.emul_work_counts(C.byref(flat), rays8.shape[0], rays8.ctypes.data_as(C.c_void_p), fetched.ctypes.data_as(C.c_void_p))
            note("work counts on %d sample bands done" % len(bands))
            n_box_c, n_tri_c = work[2] / work[0], work[3] / work[0]
            b_closest = n_box_c * 32 + n_tri_c * 36 + 24 + 16
            assert int(fetched[0]) == int(work[0]) and int(fetched[4] + fetched[8]) == int(work[6])
            # records per ray the device walks fetch (64-byte node records, 48-byte triangle records)
            nodes_c, tris_c = fetched[1] / fetched[0], fetched[2] / fetched[0]
            n_sh = max(int(fetched[4] + fetched[8]), 1)
            nodes_s, tris_s = (fetched[5] + fetched[9]) / n_sh, (fetched[6] + fetched[10]) / n_sh
            if world == 1:
                # CPU baseline: whole passes of the same frame until about 20 s of CPU time are spent
                rgb[:] = 0
                pas[:] = 0
                cnt = np.zeros(3, dtype=np.uint64)
                n_pass = 0
                tb = time.time()
                rows = H
                while True:
                    _, _, c = oscene.render(ocam, orc.cli_lights(), W, H, (0, 0, W, rows), n_pass, 1, seed=args.seed,
                                            threads=threads, rgb=rgb, passes=pas)
                    cnt += c
                    n_pass += 1
                    t_cpu = time.time() - tb
                    if t_cpu * threads >= 20.0 or n_pass >= 64:
                        break
                n_samp = int(pas.sum())
                cpu = {"value": round(float(cnt[0] + cnt[2]) / t_cpu / 1e6, 3), "unit": "Mrays/s", "cores": threads,
                       "kind": "port",
                       "value_definition": "the same rays as `value` counts (closest-hit casts + shadow casts of light samples "
                                           "with non-zero weight) per second; the CPU code casts the zero-weight ones too, "
                                           "see reference_counted_Mrays_per_s",
                       "reference_counted_Mrays_per_s": round(float(cnt[0] + cnt[1]) / t_cpu / 1e6, 3),
                       "sample": "CPU oracle (strict-IEEE restatement of the reference path), %d threads, %d full "
                                 "%dx%d passes = %d camera samples in %.2f s; extrapolates to %.0f s per 1000 spp"
                                 % (threads, n_pass, W, H, n_samp, t_cpu, t_cpu / n_samp * W * H * 1000),
                       "camera_samples_per_s": round(n_samp / t_cpu, 1),
                       "oracle_scene_build_s": round(t_oracle_build, 2)}
            scene_bytes = host.flat.n_nodes * 64 + host.n * (48 + 64)
            key = "level%d_%dx%d" % (args.level, W, H) + ("_lambert" if args.lambertian else "")
            tr = measured_traffic(key)

            variant = gpu.trace_variant()
            # a lane's walk reads 48-byte triangle records; the wave-wide leaf tests read 36 bytes per triangle
            tri_bytes = 36 if variant["leafwise"] else 48

            def price(name, cls, rays, nodes_per_ray, tris_per_ray, io_bytes, algorithmic_bytes):
                """Roofline entry of one trace-kernel class from this run's event times."""
                ms, launches = kt[cls]
                if not (ms > 0 and launches > 0 and rays > 0):
                    return None
                rays_per_launch = rays / launches
                launch_s = ms / launches * 1e-3
                loaded_per_ray = nodes_per_ray * 64 + tris_per_ray * tri_bytes + io_bytes
                loaded = rays_per_launch * loaded_per_ray / launch_s / 1e9
                e = {"kernel": name, "trace_variant": variant, "share_of_kernel_ms": round(ms / sum(v[0] for v in kt.values()), 4),
                     "rays_per_launch": round(rays_per_launch, 1), "avg_launch_ms": round(ms / launches, 4),
                     "node_records_per_ray": round(float(nodes_per_ray), 3), "tri_records_per_ray": round(float(tris_per_ray), 3),
                     "loaded_bytes_per_ray": round(float(loaded_per_ray), 1), "loaded_GBps": round(loaded, 2),
                     "frac_of_l2_gather_rate": round(loaded / L2_GATHER_PEAK_GBS, 5),
                     "frac_of_l2_aggregate": round(loaded / L2_AGGREGATE_GBS, 5),
                     "frac_of_own_microbenchmark": round(loaded / OWN_GATHER_RATE_GBS, 4)}
                if variant["leafwise"]:
                    e["triangle_count_note"] = ("triangles per ray as a lane's own walk meets them (an any-hit walk stops inside a leaf at its "
                                                "first hit); the wave-wide tests fetch whole leaves, 36 bytes per triangle, so this is a lower bound")
                if algorithmic_bytes is not None:
                    alg = rays_per_launch * algorithmic_bytes / launch_s / 1e9
                    e.update({"algorithmic_bytes_per_ray": round(float(algorithmic_bytes), 1), "algorithmic_GBps": round(alg, 2),
                              "algorithmic_over_hbm_peak": round(alg / HBM_PEAK_GBS, 4)})
                pre = "closest" if cls == 0 else "shadow"
                if tr and pre + "_bytes_per_ray" in tr:
                    g = tr[pre + "_bytes_per_ray"] * rays_per_launch / launch_s / 1e9
                    e["traffic"] = round(tr[pre + "_bytes_per_ray"] * rays_per_launch, 1)
                    e["hbm_measured"] = {"GBps": round(g, 1), "frac_of_hbm_peak": round(g / HBM_PEAK_GBS, 4),
                                         "bytes_per_ray": round(tr[pre + "_bytes_per_ray"], 1),
                                         "l2_hit_rate": round(tr.get(pre + "_l2_hit_rate", float("nan")), 4),
                                         "source": "profiles/traffic.json[%s] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                                   "command, 2*FETCH_SIZE + WRITE_SIZE) x this run's rays per second in the kernel" % key}
                    if pre + "_valu_issue_utilisation" in tr:
                        e["valu_measured"] = {"issue_utilisation": round(tr[pre + "_valu_issue_utilisation"], 4),
                                              "lane_utilisation": round(tr[pre + "_valu_lane_utilisation"], 4),
                                              "instructions_per_64_rays": round(tr[pre + "_valu_instructions_per_64_rays"], 1)}
                else:
                    e["traffic"] = None
                return e

            # closest-hit: per ray the 24-byte ray in and the 16-byte hit out; any-hit: the ray, its light distance and
            # target entry in (32 bytes), one visibility byte out
            pc = price("k_trace<closest>", 0, tally.closest, nodes_c, tris_c, 24 + 16, b_closest)
            ps = price("k_trace<shadow>", 1, tally.traced, nodes_s, tris_s, 32 + 1, None)
            entries = [e for e in (pc, ps) if e]
            if entries:
                dom = max(entries, key=lambda e: e["share_of_kernel_ms"])
                other = [e for e in entries if e is not dom]
                streams_from_hbm = scene_bytes > INFINITY_CACHE_BYTES and dom.get("algorithmic_GBps") is not None \
                    and dom["algorithmic_GBps"] <= HBM_PEAK_GBS
                if streams_from_hbm:
                    # the scene streams from HBM: the algorithmic bytes are priced against the HBM peak
                    roof = {"bound": "hbm", "achieved": dom["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(dom["algorithmic_GBps"] / HBM_PEAK_GBS, 5)}
                else:
                    # The scene (nodes + triangle records) is served by L2 / Infinity Cache: what binds is the compute
                    # units' gather path, priced with the bytes the kernel really loads.  The ceiling quoted is the
                    # hardware guide's MEASURED rate for gathers of L2-resident rows (16.8 TB/s), not a peak; the
                    # fraction of the L2's aggregate bandwidth (34.5 TB/s) is given beside it.
                    roof = {"bound": "l2", "achieved": dom["loaded_GBps"], "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s",
                            "frac": round(dom["loaded_GBps"] / L2_GATHER_PEAK_GBS, 5),
                            "peak_is": "a measured gather rate (hardware guide: 16.8-18.8 TB/s for L2-resident row gathers), not a "
                                       "hardware peak; frac_of_l2_aggregate prices the same bytes against the L2's 34.5 TB/s",
                            "peak_sources": {"hardware_guide_l2_resident_row_gather_GBps": [16800.0, 18800.0],
                                             "hardware_guide_l2_aggregate_GBps": L2_AGGREGATE_GBS,
                                             "own_microbenchmark_64B_per_lane_records_GBps": OWN_GATHER_RATE_GBS,
                                             "own_microbenchmark_source": "scripts/micro/gather_rate.hip, profiles/r01i_micro_gather_rate.txt"}}
                roof["traffic"] = dom.get("traffic")
                roof["scene_bytes"] = int(scene_bytes)
                roof["work_sample"] = "%d bands of 8 rows, 1 spp: %d closest-hit and %d traced shadow rays replayed through the lane logic" \
                    % (len(bands), int(fetched[0]), n_sh)
                roof["reference_order_tests_per_ray"] = {"closest_box": round(float(n_box_c), 3), "closest_tri": round(float(n_tri_c), 3),
                                                         "shadow_box": round(float(work[4] / max(work[1], 1)), 3),
                                                         "shadow_tri": round(float(work[5] / max(work[1], 1)), 3)}
                roof["algorithmic_note"] = ("algorithmic bytes = every box and triangle test of the reference's visit order priced at its "
                                            "record's bytes (closest-hit only: the any-hit walk stops at the first hit, so the reference's "
                                            "counts are not its work); above what reaches HBM when the scene is cache-resident, so a ratio, "
                                            "not a fraction of a roofline")
                roof.update({k: v for k, v in dom.items() if k != "traffic"})
                roof["other_trace_kernel"] = other[0] if other else None
        out["roofline"] = roof
        out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
