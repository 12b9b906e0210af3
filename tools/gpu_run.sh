#!/bin/bash
# GPU session driver: tools/gpu_run.sh <mode>...   (modes: smoke micro test testall bench benchvar sweep prof pmc)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
short() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms=%.4f achieved=%.0f GB/s frac=%.3f rows/s=%.3e parity=%s' % (d['roofline']['kernel_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['value'], d.get('parity',{}).get('bit_exact_vs_oracle')))"; }
for mode in "$@"; do
case $mode in
smoke)
  echo "== smoke =="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ;;
micro)
  echo "== microbench =="; timeout 300 ./tools/microbench > $OUT/microbench.jsonl 2>&1; tail -60 $OUT/microbench.jsonl ;;
lane)
  echo "== lane-contiguous direct loads =="; timeout 300 ./tools/microbench lane 2>&1 | tee $OUT/microbench_lane.jsonl ;;
atomics)
  echo "== lds atomic mixes =="; timeout 300 ./tools/microbench atomics 2>&1 | tee $OUT/microbench_atomics.jsonl ;;
c3)
  echo "== group-by tests + C3 =="; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_typed.py tests/test_gpu_golden.py -m gpu -x -q -k "group or golden" 2>&1 | tail -5
  for v in ${C3_VARIANTS:-"PINOT_GPU_GROUP_PRIVATE=1" "PINOT_GPU_GROUP_PRIVATE=0" "PINOT_GPU_GROUP_PACK=0" "PINOT_GPU_GROUP_WAVES=8" "PINOT_GPU_BLOCKS_PER_CU=2"}; do
    echo "-- $v"; NC=--no-check; [ "$v" = "PINOT_GPU_GROUP_PRIVATE=1" ] && NC=""; env $v timeout 900 python tools/bench_configs.py --match "C3|GROUP" --only c23 $NC 2>&1 | grep -E "C3|GROUP|Error|error" | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('   %-52s %.3f ms %6.0f GB/s exact=%s' % (d['config'], d['kernel_ms'], d['GBps'], d['bit_exact_vs_oracle']))
    except Exception: print(l.rstrip())"
  done ;;
c3sq)
  echo "== rocprof SQ counters, C3 =="; cd /tmp && export TMPDIR=/tmp
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
    rm -rf $OUT/pmc_c3; timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_c3 -o c3 -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --match "C3 SUM\(a\), MAX\(b\) GROUP" --only c23 --no-check > $OUT/pmc_c3.log 2>&1
    tail -1 $OUT/pmc_c3.log
    for f in $(find $OUT/pmc_c3 -name "*counter_collection*.csv"); do python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'scan_group' in r['Kernel_Name']:
        agg[r['Counter_Name']][r['Dispatch_Id']].append(float(r['Counter_Value']))
for c, d in agg.items():
    vals = [sum(v) for v in d.values()]
    print('   %-24s per-dispatch mean %.4g (n=%d)' % (c, sum(vals) / len(vals), len(vals)))
PY
    done
  done
  cd $GRAFT_REPO_ROOT ;;
golden)
  echo "== pytest gpu golden =="; timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -8 ;;
test)
  echo "== pytest gpu =="; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ;;
testall)
  echo "== pytest gpu (no -x) =="; timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 ;;
benchgcd0)
  echo "== bench 1B, PINOT_GPU_PLANE_GCD=0 (plain value - min plane) =="; PINOT_GPU_PLANE_GCD=0 timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_1B_plane_gcd0.json 2>/dev/null; cat $OUT/bench_1B_plane_gcd0.json | short ;;
bench)
  echo "== bench 1B =="; timeout 900 python bench.py --steps 20 --warmup 3 --extra > $OUT/bench_1B.json 2> $OUT/bench_1B.err; cat $OUT/bench_1B.json; grep extra $OUT/bench_1B.err ;;
benchvar)
  for v in ${BENCH_VARIANTS:-"PINOT_GPU_PREFETCH=1" "PINOT_GPU_PREFETCH=0" "PINOT_GPU_SCAN_PRIVATE=0" "PINOT_GPU_BLOCKS_PER_CU=8" "PINOT_GPU_BLOCKS_PER_CU=3"}; do
    echo "== bench $v =="; env $v timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --extra 2> $OUT/tmp.err | short; grep extra $OUT/tmp.err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-34s %.3f ms %6.0f GB/s' % (d['extra'], d['kernel_ms'], d['GBps']))"
  done ;;
profile)
  echo "== wave profile =="; timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-waves 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"roofline\"][\"kernel_ms\"], json.dumps(d[\"wave_profile\"]))" ;;
configs)
  echo "== all configs =="; timeout 1500 python tools/bench_configs.py --out $OUT/configs.jsonl 2>&1 | tail -30 ;;
sweep)
  for bpc in 2 3 4 5 6 8 10 12; do echo "== bench bpc=$bpc =="; PINOT_GPU_BLOCKS_PER_CU=$bpc timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | short; done ;;
c6pmc)
  echo "== rocprof pmc FETCH_SIZE / WRITE_SIZE, C6 wide group-by =="; cd /tmp && export TMPDIR=/tmp
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_c6_$ctr
    timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_c6_$ctr -o c6 -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --match "C6 SUM\(a\) GROUP BY k, f \(1M" --only c23 --no-check --no-settle > $OUT/pmc_c6_$ctr.log 2>&1
    for f in $(find $OUT/pmc_c6_$ctr -name "*counter_collection*.csv"); do python - "$f" "$ctr" <<'PY'
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "group_partition" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]:
        per[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, d in per.items():
    v = sum(d.values()) / len(d)
    print("%-60s %s mean per launch = %.1f KiB raw  (x2 on gfx950 for FETCH_SIZE streaming reads: %.3f GB; as is: %.3f GB)" % (k, sys.argv[2], v, 2 * v * 1024 / 1e9, v * 1024 / 1e9))
PY
    done
    find $OUT/pmc_c6_$ctr -name "*.csv" -size +8M -delete
  done
  cd $GRAFT_REPO_ROOT ;;
c6prof)
  echo "== rocprof kernel stats, C6 wide group-by =="; cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c6 -o c6 -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --match "${C6_MATCH:-C6 SUM\(a\) GROUP BY k, f \(1M}" --only c23 --no-check > $OUT/prof_c6.log 2>&1
  tail -2 $OUT/prof_c6.log | cut -c1-300; find $OUT/prof_c6 -name "*kernel_stats*.csv" -exec cat {} \; | head -14
  find $OUT/prof_c6 -name "*kernel_trace*.csv" -size +8M -delete
  cd $GRAFT_REPO_ROOT ;;
prof)
  echo "== rocprof =="; cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/rocprof.log 2>&1
  tail -2 $OUT/rocprof.log; find $OUT/prof -name "*kernel_stats*.csv" -exec cat {} \; | head -12
  find $OUT/prof -name "*kernel_trace*.csv" -size +8M -delete
  cd $GRAFT_REPO_ROOT ;;
sq)
  echo "== rocprof SQ counters =="; cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
  tail -2 $OUT/pmc_sq.log
  for f in $(find $OUT/pmc_sq -name "*counter_collection*.csv"); do head -1 $f; grep -E "scan_agg|scan_private" $f | head -16; done
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq2.log 2>&1
  for f in $(find $OUT/pmc_sq2 -name "*counter_collection*.csv"); do head -1 $f; grep -E "scan_agg|scan_private" $f | head -16; done
  cd $GRAFT_REPO_ROOT ;;
pmc)
  echo "== rocprof pmc =="; cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
  tail -2 $OUT/pmc_fetch.log
  for f in $(find $OUT/pmc_fetch -name "*counter_collection*.csv"); do head -1 $f; grep -E "scan_agg|scan_private" $f | head -4; done
  timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_tcc.log 2>&1
  for f in $(find $OUT/pmc_tcc -name "*counter_collection*.csv"); do head -1 $f; grep -E "scan_agg|scan_private" $f | head -4; done
  cd $GRAFT_REPO_ROOT ;;
esac
done
