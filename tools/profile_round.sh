#!/bin/bash
# tools/profile_round.sh <round> [step...]: the rocprofv3 evidence of a round, left under gpurun_out/<round>/ (copy what is to be judged into profiles/<round>/)
#   stats     kernel trace + stats of the driver's own command line (python bench.py, all variants)
#   head      HBM traffic of the headline kernel (FETCH_SIZE, TCC) -- separate --pmc passes, as MI355X_MICROARCH.md prescribes
#   c3        SQ / LDS counters and HBM traffic of the C3 group-by kernel
#   c5        HBM traffic of the C5 kernels (index_and_kernel, scan_sparse_kernel), one counter per pass
#   hist narrow lowsel    HBM traffic of scan_hist_kernel (C2b-irregular), scan_narrow_kernel (C5-scan-count), scan_private_kernel at 1 % (C2b-1pct)
#   per-variant           kernel stats of one variant per rocprofv3 run (averages not mixed across variants)
# no step = stats head c3 c5
R=${1:-r2}; shift
STEPS=${@:-stats head c3 c5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pmc() {   # pmc <tag> <kernel regexp> <bench args> <counters...>
  local tag=$1 kern=$2 args=$3; shift 3
  rm -rf $OUT/pmc_$tag
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py $args > $OUT/pmc_$tag.log 2>&1
  for f in $(find $OUT/pmc_$tag -name "*counter_collection*.csv"); do python - "$f" "$kern" "$tag" <<'PY'
import csv, sys, collections, re, json
import os
per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Kernel_Name"]):
        per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
# PMC_LAST=N: only the LAST N dispatches of every kernel count.  A bench run launches the headline query first (clock settle, warm-up,
# steps) and the named variant after it: the same kernel serves both, and a mean over all of its dispatches mixes two queries (round 3's
# "1.5x over-fetch" of the 1 % query was that mix: 52 dispatches of the 10 % headline averaged with 51 of the variant).
last = int(os.environ.get("PMC_LAST", "0"))
out = {}
for k, cs in per.items():
    out[k] = {}
    for c, d in cs.items():
        ids = sorted(d)
        if last > 0: ids = ids[-last:]
        vals = [d[i] for i in ids]
        out[k][c] = {"mean_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals), "min": min(vals), "max": max(vals), "last_n_only": last}
    print("  %-40s %s" % (k[-40:], "  ".join("%s=%.4g" % (c, v["mean_per_dispatch"]) for c, v in out[k].items())))
json.dump(out, open(sys.argv[1].rsplit("/", 1)[0] + "/../pmc_%s_summary.json" % sys.argv[3], "w"), indent=1)
PY
  done
  find $OUT/pmc_$tag -name "*.csv" -size +8M -delete
}
ONE="--steps 3 --warmup 1 --segments 1 --no-cpu-baseline"
for step in $STEPS; do
case $step in
stats)
  echo "== kernel stats, python bench.py (all variants)"
  timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
  find $OUT/stats -name "*kernel_stats*.csv" -exec cat {} \; | cut -c1-200 | head -24
  find $OUT/stats -name "*kernel_trace*.csv" -size +8M -delete ;;
head)
  echo "== headline HBM traffic"
  pmc head_fetch "scan_simple_kernel|scan_private_kernel" "$ONE --no-variants --no-clock-settle" FETCH_SIZE
  pmc head_tcc "scan_simple_kernel|scan_private_kernel" "$ONE --no-variants --no-clock-settle" TCC_HIT_sum TCC_MISS_sum ;;
c3)
  echo "== C3 group-by kernel: SQ / LDS counters, HBM traffic"
  pmc c3_sq "group_private_kernel" "$ONE --variants ^C3$" SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pmc c3_sq2 "group_private_kernel" "$ONE --variants ^C3$" SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU
  pmc c3_fetch "group_private_kernel" "$ONE --variants ^C3$" FETCH_SIZE ;;
c3lds)
  echo "== C3 group-by kernel: LDS conflict counters only"
  pmc c3_sq "group_private_kernel" "$ONE --variants ^C3$" SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ;;
c5)
  echo "== C5 kernels: HBM traffic (one counter per pass)"
  pmc c5_fetch "index_and|scan_private_kernel|scan_sparse" "--steps 2 --warmup 1 --segments 1 --no-cpu-baseline --rows 100000 --rows-c5 1000000000 --variants ^C5-dense$" FETCH_SIZE
  pmc c5s_fetch "index_and|scan_private_kernel|scan_sparse" "--steps 2 --warmup 1 --segments 1 --no-cpu-baseline --rows 100000 --rows-c5 1000000000 --variants ^C5-sparse$" FETCH_SIZE ;;
c5s_sq)
  echo "== what index_and_kernel's wavefronts do with their time (C5-sparse-count: the AND alone)"
  C5S="--steps 2 --warmup 1 --segments 1 --no-cpu-baseline --rows 100000 --rows-c5 1000000000 --variants ^C5-sparse-count$"
  pmc c5s_sq "index_and_kernel" "$C5S" SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
  pmc c5s_sq2 "index_and_kernel" "$C5S" SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
  pmc c5s_sq3 "index_and_kernel" "$C5S" GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH ;;
hist)
  echo "== scan_hist_kernel (C2b-irregular): HBM traffic"
  pmc hist_fetch "scan_hist_kernel" "$ONE --variants ^C2b-irregular$" FETCH_SIZE ;;
narrow)
  echo "== scan_narrow_kernel (C5-scan-count): HBM traffic"
  pmc narrow_fetch "scan_narrow" "--steps 2 --warmup 1 --segments 1 --no-cpu-baseline --rows 100000 --rows-c5 1000000000 --variants ^C5-scan-count$" FETCH_SIZE ;;
lowsel)
  echo "== scan_private_kernel at 1 % (C2b-1pct): HBM traffic"
  PMC_LAST=40 pmc lowsel_fetch "scan_simple_kernel|scan_private_kernel" "$ONE --no-clock-settle --variants ^C2b-1pct$" FETCH_SIZE ;;
per-variant)
  echo "== kernel stats, one variant per run"
  for v in C2b-irregular C2b-1pct C2a-affine C3 C3-filter COUNT-filter; do
    rm -rf $OUT/stats_$v
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$v -o v -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --segments 1 --no-cpu-baseline --variants "^$v\$" > $OUT/stats_$v.json 2> $OUT/stats_$v.err
    echo "-- $v"; find $OUT/stats_$v -name "*kernel_stats*.csv" -exec cat {} \; | cut -c1-160 | head -4
    find $OUT/stats_$v -name "*kernel_trace*.csv" -delete
  done ;;
*) echo "unknown step $step" ;;
esac
done
