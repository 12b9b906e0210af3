#!/usr/bin/env python3
"""The headline kernel's timed dispatches picked out of a rocprofv3 kernel trace of `python bench.py`.

bench.py launches the headline query 48 (clock settle) + warmup x segments times before its `steps x segments` timed launches, then the
cold-launch samples, the overlapped step and the variants -- the same kernel serves several of those.  This takes the kernel's dispatches
[settle + warmup x segments, + steps x segments) in dispatch order and prints their mean duration and what it is of 8 TB/s on the
algorithmic bytes (3.375 B per row), so that the figure bench.py measures with HIP events can be held against the profiler's.
    python tools/headline_from_trace.py <kernel_trace.csv> [--kernel scan_simple_kernel] [--settle 48] [--warmup 3] [--steps 20] [--segments 8] [--bytes 3375000000]
"""
import argparse
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--kernel", default="scan_simple_kernel")
    ap.add_argument("--settle", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--segments", type=int, default=8)
    ap.add_argument("--bytes", type=float, default=3.375e9)
    args = ap.parse_args()
    rows = []
    with open(args.trace, newline="") as f:
        for r in csv.DictReader(f):
            if args.kernel in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    first = args.settle + args.warmup * args.segments
    timed = [d for _, d in rows[first:first + args.steps * args.segments]]
    avg_us = sum(timed) / len(timed) / 1e3
    print(json.dumps({"kernel": args.kernel, "dispatches_of_the_kernel_in_the_trace": len(rows), "first_timed_dispatch": first, "timed_dispatches": len(timed),
                      "avg_us": avg_us, "min_us": min(timed) / 1e3, "max_us": max(timed) / 1e3, "algorithmic_bytes_per_launch": args.bytes,
                      "achieved_GBps": args.bytes / avg_us / 1e3, "frac_of_8TBps": args.bytes / avg_us / 1e3 / 8000.0}))


if __name__ == "__main__":
    main()
