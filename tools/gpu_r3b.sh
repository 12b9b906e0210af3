#!/bin/bash
# Round-3 probes: tools/gpu_r3b.sh <step>...   (everything lands under gpurun_out/r3/)
# Steps that compare two BUILDS expect the variant library under tools/ (git-ignored, shipped by gpurun); build it first with
#   make -C pinot_amd/csrc variant NAME=simple6 UNIT=pg_unit_scan_simple DEFS=-DPG_SIMPLE_WAVES=6     (step `simple`)
#   make -C pinot_amd/csrc variant NAME=nopair  UNIT=pg_unit_scan_simple DEFS=-DPG_SIMPLE_PAIR=0      (step `pair`: the pair path itself was removed after the measurement)
#   make -C pinot_amd/csrc variant NAME=sparse4 UNIT=pg_unit_scan_sparse DEFS=-DPG_SPARSE_WAVES=4     (step `slots`; the default is 4 again: use =5 for the other side)
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $OUT
for step in "$@"; do
case $step in
leap)
  echo "== A/B: two-leaf AND counted on the device vs not; sparse walk thresholds =="
  timeout 600 python tools/ab_r3.py --match "AND2|C2b-1pct|C2b-0.1pct" --settings default,leap0,sparse0,sparse64 --check > $OUT/ab_leap_sparse.jsonl 2> $OUT/ab_leap_sparse.err
  tail -3 $OUT/ab_leap_sparse.err
  python - <<'PY'
import json, os
for l in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3/ab_leap_sparse.jsonl")):
    if l.startswith("{"):
        r = json.loads(l)
        print("   %-10s %-12s %-24s kernel %.4f all %.4f wall %.4f entries %d exact=%s same=%s oracle=%s" % (r["setting"], r["query"], r["kernel"], r["kernel_ms"], r["all_kernels_ms"],
              r["wall_ms_untimed"], r["entries_in_filter"], r["entries_exact"], r["same_as_first_setting"], r.get("bit_exact_vs_oracle")))
PY
  ;;
batchtrace)
  echo "== host phases of pg_execute_batch (C1x64) =="
  PINOT_GPU_BATCH_TRACE=1 timeout 600 python bench.py --steps 2 --warmup 1 --segments 1 --rows 1000000 --no-cpu-baseline --no-clock-settle --variants "^C1x64" > $OUT/batch_trace.json 2> $OUT/batch_trace.err
  grep -c "pg_execute_batch" $OUT/batch_trace.err
  grep "pg_execute_batch\|run_deferred" $OUT/batch_trace.err | tail -12 ;;
batchtest)
  echo "== pytest: batch, planes, fold, concurrency =="; timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_planes.py tests/test_gpu_fold.py tests/test_gpu_hist.py -m gpu -x -q 2>&1 | tail -4
  timeout 300 python tools/stress_concurrency.py 2>&1 | tail -3 ;;
batchbench)
  echo "== C1x64 variants (driver-style bench line, only these variants) =="
  timeout 600 python bench.py --steps 3 --warmup 1 --segments 1 --rows 1000000 --no-cpu-baseline --no-clock-settle --variants "^C1x64" > $OUT/bench_c1x64.json 2> $OUT/bench_c1x64.err
  python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3/bench_c1x64.json")))
for v in d.get("variants", []):
    print("   %-18s kernel %.4f  %s  exact=%s" % (v["id"], v["kernel_ms"], "  ".join("%s %.4f (min %.4f)" % (m, x["wall_ms"], x["wall_ms_min"]) for m, x in v["modes"].items()), v["bit_exact_vs_oracle"]))
PY
  ;;
kernarg)
  echo "== C1 probe with kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) =="
  for v in 0 1; do
    HIP_FORCE_DEV_KERNARG=$v timeout 300 python tools/c1_probe.py --sizes 2048,1000000,10000000 > $OUT/c1_probe_kernarg$v.jsonl 2> $OUT/c1_probe_kernarg$v.err
    echo "-- HIP_FORCE_DEV_KERNARG=$v"; python - $OUT/c1_probe_kernarg$v.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        print("   %9d %-12s kernel %6.2f us (min %6.2f)  wall %6.2f us (min %6.2f)" % (r["rows"], r["query"], r["kernel_us"], r["kernel_us_min"], r["wall_us"], r["wall_us_min"]))
PY
  done ;;
onecounter)
  echo "== fold with one arrival counter: tests, then C1 probe default / onecounter =="
  PINOT_GPU_FOLD_ONE_COUNTER=1 timeout 600 python -m pytest tests/test_gpu_fold.py tests/test_gpu_batch.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
  timeout 300 python tools/c1_probe.py --sizes 2048,200000,1000000,4000000,10000000,40000000 --settings default,onecounter > $OUT/c1_probe_onecounter.jsonl 2> $OUT/c1_probe_onecounter.err
  tail -2 $OUT/c1_probe_onecounter.err
  python - $OUT/c1_probe_onecounter.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        print("   %-10s %9d %-12s kernel %6.2f us (min %6.2f)  wall %6.2f us (min %6.2f) exact=%s" % (r["setting"], r["rows"], r["query"], r["kernel_us"], r["kernel_us_min"], r["wall_us"], r["wall_us_min"], r.get("bit_exact_vs_oracle")))
PY
  ;;
simple)
  echo "== scan_simple_kernel: parity tests, then A/B against scan_private_kernel (and the six-waves build) =="
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fold.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py tests/test_gpu_nulls.py -m gpu -x -q 2>&1 | tail -4
  show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        print("   %-10s %-14s %-22s kernel %.4f all %.4f wall %.4f same=%s oracle=%s" % (r["setting"], r["query"], r["kernel"], r["kernel_ms"], r["all_kernels_ms"], r["wall_ms_untimed"], r["same_as_first_setting"], r.get("bit_exact_vs_oracle")))
PY
  }
  M="C2b-10pct|C2b-3pct|C2b-1pct|COUNT-filter|MINMAXAVG|C1-dict-sum"
  timeout 600 python tools/ab_r3.py --match "$M" --settings default,simple0 --check > $OUT/ab_simple.jsonl 2> $OUT/ab_simple.err; tail -2 $OUT/ab_simple.err; show $OUT/ab_simple.jsonl
  echo "-- six waves per SIMD (80 VGPRs, 26 spilled)"
  PINOT_GPU_LIB=$GRAFT_REPO_ROOT/tools/libpinot_gpu_simple6.so timeout 600 python tools/ab_r3.py --match "$M" --settings default > $OUT/ab_simple6.jsonl 2> $OUT/ab_simple6.err; tail -2 $OUT/ab_simple6.err; show $OUT/ab_simple6.jsonl
  ;;
pair)
  echo "== scan_simple_kernel: two tiles per iteration vs one (PINOT_GPU_LIB=tools/libpinot_gpu_nopair.so) =="
  timeout 600 python -m pytest tests/test_gpu_scan_simple.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
  show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        print("   %-10s %-14s %-22s kernel %.4f all %.4f wall %.4f (min %.4f) same=%s oracle=%s" % (r["setting"], r["query"], r["kernel"], r["kernel_ms"], r["all_kernels_ms"], r["wall_ms_untimed"], r["wall_ms_untimed_min"], r["same_as_first_setting"], r.get("bit_exact_vs_oracle")))
PY
  }
  M="C2b-10pct|C2b-3pct|C2b-1pct|COUNT-filter|MINMAXAVG|C1-dict-sum"
  for round in 1 2; do
    echo "-- pair (round $round)"; timeout 600 python tools/ab_r3.py --match "$M" --settings default --check > $OUT/ab_pair_$round.jsonl 2> $OUT/ab_pair.err; tail -2 $OUT/ab_pair.err; show $OUT/ab_pair_$round.jsonl
    echo "-- no pair (round $round)"; PINOT_GPU_LIB=$GRAFT_REPO_ROOT/tools/libpinot_gpu_nopair.so timeout 600 python tools/ab_r3.py --match "$M" --settings default > $OUT/ab_nopair_$round.jsonl 2> $OUT/ab_nopair.err; tail -2 $OUT/ab_nopair.err; show $OUT/ab_nopair_$round.jsonl
  done ;;
soak)
  echo "== fuzz soak over other seeds =="
  for base in ${SOAK_BASES:-100 200 300 400 500 600}; do
    echo "-- seed base $base"; PINOT_FUZZ_SEED_BASE=$base timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
  done ;;
slots)
  echo "== one-slot typed / sparse kernels: tests, C7 + C1 (typed), C5 five vs four waves (sparse) =="
  timeout 900 python -m pytest tests/test_gpu_typed.py tests/test_gpu_index_and.py tests/test_gpu_planes.py tests/test_gpu_fold.py -m gpu -x -q 2>&1 | tail -3
  timeout 600 python tools/bench_configs.py --match "C7" > $OUT/configs_C7_one_slot.jsonl 2> $OUT/configs_C7_one_slot.err; tail -2 $OUT/configs_C7_one_slot.err
  python - $OUT/configs_C7_one_slot.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("   %-70s %-28s %.4f ms %6.0f GB/s exact=%s" % (d["config"][:70], d.get("kernel", ""), d["kernel_ms"], d["GBps"], d.get("bit_exact_vs_oracle")))
PY
  timeout 300 python tools/c1_probe.py --sizes 10000000,40000000 > $OUT/c1_probe_one_slot.jsonl 2> $OUT/c1_probe_one_slot.err
  python - $OUT/c1_probe_one_slot.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        print("   %9d %-12s %-26s kernel %6.2f us (min %6.2f)  wall %6.2f us" % (r["rows"], r["query"], r["kernel"], r["kernel_us"], r["kernel_us_min"], r["wall_us"]))
PY
  show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        print("   %-10s %-16s %-22s kernel %.4f all %.4f wall %.4f same=%s oracle=%s" % (r["setting"], r["query"], r["kernel"], r["kernel_ms"], r["all_kernels_ms"], r["wall_ms_untimed"], r["same_as_first_setting"], r.get("bit_exact_vs_oracle")))
PY
  }
  echo "-- C5, scan_sparse_kernel<1> at five waves per SIMD"; timeout 900 python tools/ab_r3.py --c5 --match "C5" --settings default --check > $OUT/ab_sparse5.jsonl 2> $OUT/ab_sparse5.err; tail -2 $OUT/ab_sparse5.err; show $OUT/ab_sparse5.jsonl
  echo "-- four waves"; PINOT_GPU_LIB=$GRAFT_REPO_ROOT/tools/libpinot_gpu_sparse4.so timeout 900 python tools/ab_r3.py --c5 --match "C5" --settings default > $OUT/ab_sparse4.jsonl 2> $OUT/ab_sparse4.err; tail -2 $OUT/ab_sparse4.err; show $OUT/ab_sparse4.jsonl
  ;;
twoslots)
  echo "== scan_private_kernel<2> vs <4> for two aggregated columns; tests =="
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_index_and.py -m gpu -x -q 2>&1 | tail -3
  timeout 600 python tools/ab_r3.py --match "TWOCOL" --settings default --check > $OUT/ab_twoslots.jsonl 2> $OUT/ab_twoslots.err; tail -2 $OUT/ab_twoslots.err
  python - $OUT/ab_twoslots.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        print("   %-10s %-16s %-22s kernel %.4f all %.4f wall %.4f same=%s oracle=%s" % (r["setting"], r["query"], r["kernel"], r["kernel_ms"], r["all_kernels_ms"], r["wall_ms_untimed"], r["same_as_first_setting"], r.get("bit_exact_vs_oracle")))
PY
  ;;
*) echo "unknown step $step" ;;
esac
done
