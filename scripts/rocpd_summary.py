#!/usr/bin/env python3
"""Turns rocprofv3 rocpd databases (*.db) into short text summaries.

    python scripts/rocpd_summary.py <dir-with-db-files> [--kernels k_] > profiles/<name>.txt

For every database found below the directory: the kernel-trace statistics (calls, total and
average duration per kernel) and, where counters were collected, the per-kernel sum of every
counter over its dispatches.  Only kernels whose name contains one of the --kernels
substrings are listed (default: this library's kernels).
"""
import argparse
import glob
import os
import sqlite3


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 70 else name[:67] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--kernels", default="k_")
    args = ap.parse_args()
    keys = args.kernels.split(",")
    for path in sorted(glob.glob(os.path.join(args.root, "**", "*.db"), recursive=True)):
        db = sqlite3.connect(path)
        cur = db.cursor()
        print("==", os.path.relpath(path, args.root))
        try:
            rows = cur.execute(
                "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                "from kernels group by name order by sum(duration) desc").fetchall()
        except sqlite3.Error:
            rows = []
        total = sum(r[2] for r in rows) or 1
        if rows:
            print("  kernel-trace: %-62s %6s %12s %12s %12s %12s %6s" % ("name", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
        for name, calls, tot, avg, mn, mx in rows:
            if any(k in name for k in keys):
                print("  %-76s %6d %12.1f %12.2f %12.2f %12.2f %6.1f" % (short(name), calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
        try:
            rows = cur.execute(
                "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
                "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        except sqlite3.Error:
            rows = []
        last = None
        for kname, cname, val, nd in rows:
            if not any(k in kname for k in keys):
                continue
            if kname != last:
                print("  counters: %s  (%d dispatches)" % (short(kname), nd))
                last = kname
            print("      %-36s %20.0f" % (cname, val))
        db.close()


if __name__ == "__main__":
    main()
