#!/usr/bin/env python3
"""Fold the rocprofv3 counter CSVs that tools/gpu_run.sh (modes pmc, sq) left under gpurun_out/ into profiles/<round>/pmc_summary_bench_1B.json
and refresh profiles/traffic.json (the per-launch HBM bytes bench.py reports as roofline.traffic).

usage: python tools/summarize_pmc.py r1
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE is reported in KiB and under-counts wide streaming reads by 2x on gfx950
(-> doubled); TCC_MISS_sum x 128 B is the cross-check."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("scan_private_kernel", "scan_agg_kernel")


def fold(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    for r in csv.DictReader(open(path)):
        k = next((k for k in KERNELS if k in r["Kernel_Name"]), None)
        if k is None:
            continue
        per[(k, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        meta[k] = {"kernel_name": r["Kernel_Name"], "grid": r["Grid_Size"], "workgroup": r["Workgroup_Size"], "vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"]}
    out = collections.defaultdict(dict)
    for (k, c), d in per.items():
        out[k][c] = sum(d.values()) / len(d)
        out[k]["_dispatches"] = len(d)
    return out, meta


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r1"
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    groups, metas = {}, {}
    for name in ("pmc_fetch", "pmc_tcc", "pmc_sq", "pmc_sq2"):
        files = glob.glob(os.path.join(ROOT, "gpurun_out", name, "*counter_collection*.csv"))
        if not files:
            continue
        groups[name], m = fold(files[0])
        metas.update(m)
        shutil.copy(files[0], os.path.join(dst, "rocprofv3_%s_bench_1B.csv" % name))
    kernel = next((k for k in KERNELS if any(k in g for g in groups.values())), None)
    if kernel is None:
        sys.exit("no counter files under gpurun_out/")
    summary = {"command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline (one pass per counter group)",
               "kernel": metas[kernel], "workload": "C2b 1B rows, SELECT SUM(v) WHERE f < 100",
               "counters_avg_per_launch": {g: v.get(kernel, {}) for g, v in groups.items()}}
    fetch = groups.get("pmc_fetch", {}).get(kernel, {}).get("FETCH_SIZE")
    miss = groups.get("pmc_tcc", {}).get(kernel, {}).get("TCC_MISS_sum")
    if fetch is not None:
        traffic = fetch * 1024 * 2
        summary["hbm_traffic_bytes_per_launch"] = {
            "FETCH_SIZE_KiB_reported": fetch, "FETCH_SIZE_bytes_corrected_x2": traffic,
            "note": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read -> doubled",
            "TCC_MISS_sum_x128B": miss * 128 if miss is not None else None, "algorithmic_bytes": 3375000000}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        t = json.load(open(tpath)) if os.path.exists(tpath) else {}
        t[kernel] = {"workload_rows": 1000000000, "bytes_per_launch": traffic,
                     "source": "profiles/%s/pmc_summary_bench_1B.json (rocprofv3 --pmc FETCH_SIZE, doubled per MI355X_MICROARCH.md; TCC_MISS_sum x 128 B = %s)" % (rnd, miss * 128 if miss else None)}
        json.dump(t, open(tpath, "w"), indent=1)
    json.dump(summary, open(os.path.join(dst, "pmc_summary_bench_1B.json"), "w"), indent=1)
    stats = glob.glob(os.path.join(ROOT, "gpurun_out", "prof", "*kernel_stats*.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "rocprofv3_kernel_stats_bench_1B.csv"))
    print(json.dumps(summary.get("hbm_traffic_bytes_per_launch"), indent=1))


if __name__ == "__main__":
    main()
