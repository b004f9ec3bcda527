#!/usr/bin/env python3
"""Derives profiles/traffic.json from a scripts/profile_round.sh output directory.

    python scripts/make_traffic_json.py <dir> [<workload key> [<traffic.json to merge into>]] > new.json

The file holds one entry per workload ("level8_1920x1080", "level10_3840x2160", ...), which is
how bench.py looks its own workload up.

HBM-side bytes of the closest-hit trace kernel per ray = (2 * FETCH_SIZE + WRITE_SIZE) / rays,
where FETCH_SIZE and WRITE_SIZE are the rocprofv3 counters (kilobytes) summed over every
dispatch of k_trace<1, *> in their own PMC passes, FETCH_SIZE doubled as the gfx950 correction for
16-byte-per-lane loads prescribes, and rays = closest rays per step * (steps + warmup) of the
profiled bench command (taken from its JSON line).
"""
import glob
import json
import os
import sqlite3
import sys


def counter_sum(root, counter, kernel):
    for path in glob.glob(os.path.join(root, "pmc*", "*.db")):
        db = sqlite3.connect(path)
        try:
            row = db.execute("select sum(value), count(distinct dispatch_id) from counters_collection "
                             "where counter_name=? and kernel_name like ?", (counter, "%" + kernel + "%")).fetchone()
        except sqlite3.Error:
            row = None
        db.close()
        if row and row[0]:
            return float(row[0]), int(row[1])
    return None, 0


def main():
    root = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else "level8_1920x1080"   # workload the passes were run on
    merge_into = sys.argv[3] if len(sys.argv) > 3 else None            # existing traffic.json to update
    # the PMC passes run `bench.py --steps 2 --warmup 1`; their own bench line gives the ray counts
    name = "bench_line_under_pmc.json" if os.path.exists(os.path.join(root, "bench_line_under_pmc.json")) else "bench_line_under_profiler.json"
    line = json.loads(open(os.path.join(root, name)).read().strip().splitlines()[-1])
    steps_total = line["steps"] + line["warmup"]
    out = {"source": os.path.basename(root.rstrip("/")), "bench_line": {k: line[k] for k in ("value", "ms_per_step", "steps", "warmup")}}
    for name, kernel, count_key in (("closest", "k_trace<1", "closest_rays"), ("shadow", "k_trace<2", "shadow_rays")):
        # shadow rays: the traced ones (zero-weight light samples are counted but not traced)
        rays = line.get(count_key + "_traced", line[count_key]) / line["steps"] * steps_total
        fetch_kb, nd = counter_sum(root, "FETCH_SIZE", kernel)
        write_kb, _ = counter_sum(root, "WRITE_SIZE", kernel)
        hit, _ = counter_sum(root, "TCC_HIT_sum", kernel)
        miss, _ = counter_sum(root, "TCC_MISS_sum", kernel)
        if fetch_kb is None or write_kb is None:
            continue
        out[name + "_dispatches"] = nd
        out[name + "_rays"] = rays
        out[name + "_fetch_bytes_per_ray_raw"] = fetch_kb * 1024.0 / rays
        out[name + "_write_bytes_per_ray"] = write_kb * 1024.0 / rays
        out[name + "_bytes_per_ray"] = (2.0 * fetch_kb + write_kb) * 1024.0 / rays
        if hit and miss:
            out[name + "_l2_hit_rate"] = hit / (hit + miss)
        # vector-instruction issue: a gfx950 SIMD is 32 lanes wide and issues a wave64 instruction over
        # 2 cycles (a single wave's own stream: one instruction per 4); 1024 SIMDs; the kernel's cycles
        # are GRBM_GUI_ACTIVE summed over the 8 XCDs, divided by 8
        insts, _ = counter_sum(root, "SQ_INSTS_VALU", kernel)
        lanes, _ = counter_sum(root, "SQ_THREAD_CYCLES_VALU", kernel)
        active, _ = counter_sum(root, "GRBM_GUI_ACTIVE", kernel)
        if insts and active:
            out[name + "_valu_issue_utilisation"] = insts * 2.0 / (1024.0 * active / 8.0)
            out[name + "_valu_instructions_per_64_rays"] = insts / (rays / 64.0)
        if insts and lanes:
            out[name + "_valu_lane_utilisation"] = lanes / insts / 64.0
    doc = {"configs": {}}
    if merge_into and os.path.exists(merge_into):
        doc = json.load(open(merge_into))
    doc.setdefault("configs", {})[key] = out
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
