#!/bin/bash
# tools/gpu_r6.sh <tag> <step...>: round-6 GPU sessions in named steps; everything lands under gpurun_out/<tag>/
#   ia_tests    the suites that reach index_and_kernel (its own tests, the tiny grid, the fuzz, the kernel-coverage gate, the goldens)
#   c5_ab       C5 sparse / dense on resident 1 B-row segments: index_and_kernel's grid -- one wave per window, the persistent default, 8 / 12 / 20 waves per CU
#   not_trace   rocprofv3 --kernel-trace --stats of AND-NOT-scan at 1 B rows: the episode pass per kernel
#   c5_pmc      FETCH_SIZE of the C5 kernels (tools/profile_round.sh c5)
#   rank        tools/rank_image_probe.py at 1 B rows (raw DOUBLE, 40-bit LONG)
#   suite       the whole GPU suite
#   bench       python bench.py (the driver's line)
cd $GRAFT_REPO_ROOT; TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for step in "$@"; do
echo "=== $step"; t0=$(date +%s)
case $step in
ia_tests)
  timeout 1500 python -m pytest tests/test_gpu_index_and.py tests/test_gpu_tiny_grid.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py tests/test_gpu_kernel_coverage.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/ia_tests.txt ;;
c5_ab)
  timeout 1500 python tools/ab_r6.py c5 --steps 20 --out $OUT/c5_ab.jsonl --settings "PINOT_GPU_INDEX_AND_WAVES=-1;;PINOT_GPU_INDEX_AND_WAVES=8;PINOT_GPU_INDEX_AND_WAVES=12;PINOT_GPU_INDEX_AND_WAVES=20;PINOT_GPU_INDEX_AND_WAVES=-1;" 2> $OUT/c5_ab.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%-16s %-34s kernel %.4f all %.4f host %.4f untimed %.4f frac %.3f exact %s' % (r['query'], r['setting'], r['kernel_ms'], r['all_kernels_ms'], r['host_clock_ms'], r.get('host_clock_untimed_ms') or 0, r['frac_all_kernels'] or 0, r['exact']))" ;;
c5_ab2)
  timeout 1500 python tools/ab_r6.py c5 --steps 20 --out $OUT/c5_ab2.jsonl --settings "${C5_SETTINGS:-PINOT_GPU_INDEX_AND_WAVES=-1;PINOT_GPU_INDEX_AND_WAVES=-2;PINOT_GPU_INDEX_AND_WAVES=-3;PINOT_GPU_INDEX_AND_WAVES=-4;PINOT_GPU_INDEX_AND_WAVES=0;PINOT_GPU_INDEX_AND_WAVES=-1;PINOT_GPU_INDEX_AND_WAVES=-2}" 2> $OUT/c5_ab2.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%-16s %-34s kernel %.4f all %.4f host %.4f untimed %.4f frac %.3f exact %s' % (r['query'], r['setting'], r['kernel_ms'], r['all_kernels_ms'], r['host_clock_ms'], r.get('host_clock_untimed_ms') or 0, r['frac_all_kernels'] or 0, r['exact']))" ;;
c5s_sq)
  bash tools/profile_round.sh $TAG c5s_sq 2>&1 | tail -12 ;;
fsm_tests)
  timeout 1800 python -m pytest tests/test_gpu_filter_stats.py tests/test_gpu_kernel_coverage.py -m gpu -x -q 2>&1 | tail -12 | tee $OUT/fsm_tests.txt ;;
stats_flag)
  timeout 900 python -m pytest tests/test_gpu_filter_stats.py -m gpu -x -q -k "upper_bound" 2>&1 | tail -8 | tee $OUT/stats_flag_tests.txt ;;
c3_libs)
  # C3 on resident 1 B-row columns: the default build, then A/B builds of the two LDS-table group-by units (tools/make_variant.sh)
  for spec in ${C3_LIBS:-"default:;PINOT_GPU_GROUP_WAVES=8" "c3_old:" "c3_w5:PINOT_GPU_GROUP_WAVES=10;PINOT_GPU_GROUP_WAVES=5" "c3_w6:PINOT_GPU_GROUP_WAVES=8"}; do
    lib=${spec%%:*}; settings=${spec#*:}
    echo "-- build $lib"
    if [ "$lib" = default ]; then unset PINOT_GPU_LIB; else export PINOT_GPU_LIB=$GRAFT_REPO_ROOT/tools/libpinot_gpu_$lib.so; fi
    timeout 900 python tools/ab_r6.py c3 --steps 20 --out $OUT/c3_$lib.jsonl --settings "$settings" 2> $OUT/c3_$lib.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%-14s %-62s kernel %.4f all %.4f host %.4f untimed %.4f frac %.3f exact %s' % (r['query'], r['setting'], r['kernel_ms'], r['all_kernels_ms'], r['host_clock_ms'], r.get('host_clock_untimed_ms') or 0, r['frac_all_kernels'] or 0, r['exact']))"
  done; unset PINOT_GPU_LIB ;;
spill_ab)
  # kernels that spill at four waves per SIMD against builds bounded at three (no spills): the four-slot typed scans, the scan with the transducer inside
  for spec in "default" "typed3"; do
    if [ "$spec" = default ]; then unset PINOT_GPU_LIB; else export PINOT_GPU_LIB=$GRAFT_REPO_ROOT/tools/libpinot_gpu_$spec.so; fi
    echo "-- typed, build $spec"
    timeout 900 python tools/ab_r6.py typed --rows 250000000 --steps 20 --no-check --out $OUT/spill_typed_$spec.jsonl --settings "" 2> $OUT/spill_typed_$spec.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%-26s kernel %-30s %.4f all %.4f frac %.3f' % (r['query'], r['kernel'], r['kernel_ms'], r['all_kernels_ms'], r['frac_all_kernels'] or 0))"
  done
  for spec in "default" "fsm3"; do
    if [ "$spec" = default ]; then unset PINOT_GPU_LIB; else export PINOT_GPU_LIB=$GRAFT_REPO_ROOT/tools/libpinot_gpu_$spec.so; fi
    echo "-- AND3-scan, build $spec"
    timeout 900 python tools/ab_r6.py not --steps 20 --match AND3 --no-check --out $OUT/spill_and3_$spec.jsonl --settings "" 2> $OUT/spill_and3_$spec.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%-26s kernel %-30s %.4f all %.4f frac %.3f entries_exact %s' % (r['query'], r['kernel'], r['kernel_ms'], r['all_kernels_ms'], r['frac_all_kernels'] or 0, r['entries_exact']))"
  done; unset PINOT_GPU_LIB ;;
batch_tests)
  timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_index_and.py tests/test_gpu_kernel_coverage.py -m gpu -x -q 2>&1 | tail -12 | tee $OUT/batch_tests.txt ;;
c5x64)
  timeout 1500 python bench.py --steps 3 --warmup 1 --segments 1 --rows 100000 --variants "^C5x64" > $OUT/c5x64_line.json 2> $OUT/c5x64.err; tail -3 $OUT/c5x64.err
  python - <<'PY'
import json
for v in json.load(open("gpurun_out/bench_variants.json")):
    if v["id"].startswith("C5x64"):
        print(v["id"], "kernel", v.get("kernel"), "exact", v["bit_exact_vs_oracle"], {m: (round(x["wall_ms"], 4), round(x.get("kernel_ms", 0) or 0, 4)) for m, x in v["modes"].items()})
PY
  cp gpurun_out/bench_variants.json $OUT/c5x64_variants.json ;;
c3_ab)
  timeout 1500 python tools/ab_r6.py c3 --steps 20 --out $OUT/c3_ab.jsonl --settings "${C3_SETTINGS:-}" 2> $OUT/c3_ab.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%-16s %-34s kernel %.4f all %.4f host %.4f untimed %.4f frac %.3f exact %s' % (r['query'], r['setting'], r['kernel_ms'], r['all_kernels_ms'], r['host_clock_ms'], r.get('host_clock_untimed_ms') or 0, r['frac_all_kernels'] or 0, r['exact']))" ;;
not_trace)
  rm -rf $OUT/not_trace
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/not_trace -o t -- python $GRAFT_REPO_ROOT/tools/ab_r6.py not --steps 10 --warmup 10 --match AND-NOT --no-check > $GRAFT_REPO_ROOT/$OUT/not_trace.jsonl 2> $GRAFT_REPO_ROOT/$OUT/not_trace.err)
  find $OUT/not_trace -name "*kernel_stats*.csv" -exec cat {} \; | cut -c1-220 | head -24 | tee $OUT/not_trace_kernel_stats.csv
  find $OUT/not_trace -name "*kernel_trace*.csv" -size +8M -delete
  PINOT_GPU_FSM_TRACE=1 timeout 900 python tools/ab_r6.py not --steps 3 --warmup 2 --match AND-NOT --no-check 2>&1 | grep -i "fsm\|episode" | tail -8 | tee $OUT/not_fsm_trace.txt ;;
c5_pmc)
  bash tools/profile_round.sh $TAG c5 2>&1 | tail -12 ;;
rank)
  timeout 1500 python tools/rank_image_probe.py --rows ${RANK_ROWS:-1000000000} > $OUT/rank_image.jsonl 2> $OUT/rank_image.err; cat $OUT/rank_image.jsonl; grep "rank image" $OUT/rank_image.err ;;
suite)
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/gpu_suite.txt ;;
bench)
  timeout 1500 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; tail -c 3600 $OUT/bench_line.json; cp gpurun_out/bench_variants.json $OUT/bench_variants.json 2>/dev/null; cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null ;;
*) echo "unknown step $step" ;;
esac
echo "--- $step took $(( $(date +%s) - t0 )) s"
done
