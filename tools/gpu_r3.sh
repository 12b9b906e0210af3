#!/bin/bash
# Round-3 GPU sessions: tools/gpu_r3.sh <step>...   (everything lands under gpurun_out/r3/)
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $OUT
summ() { python - "$1" <<'PY'
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["query"], []).append(r)
for q, rs in by.items():
    print(q, "docs", rs[0].get("docs_matched"))
    for r in rs:
        print("   %-18s %-28s kernel %.4f all %.4f wall_timed %.4f wall_untimed %.4f (min %.4f) same=%s oracle=%s" % (r["setting"], r.get("kernel"), r["kernel_ms"], r["all_kernels_ms"],
              r["wall_ms_timed"], r["wall_ms_untimed"], r["wall_ms_untimed_min"], r["same_as_first_setting"], r.get("bit_exact_vs_oracle")))
PY
}
for step in "$@"; do
case $step in
test)
  echo "== pytest -m gpu =="; timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log ;;
testfold)
  echo "== pytest fold =="; timeout 900 python -m pytest tests/test_gpu_fold.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_fold.log 2>&1; tail -6 $OUT/pytest_fold.log ;;
ab)
  echo "== A/B =="; timeout 1200 python tools/ab_r3.py ${AB_ARGS:---c5 --check} > $OUT/ab.jsonl 2> $OUT/ab.err; tail -3 $OUT/ab.err; summ $OUT/ab.jsonl ;;
bench)
  echo "== bench.py (driver command line) =="; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
  python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3/bench.json")))
print("value %.4e rows/s  ms_per_step %.4f  kernel_ms %.4f frac %.4f  whole-step GB/s %.0f  parity %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["hbm_GBps_whole_step"], d.get("parity", {}).get("bit_exact_vs_oracle")))
for v in d.get("variants", []):
    print("   %-18s %-26s kernel %.4f all %.4f wall %.4f frac %.3f fdom %s exact=%s" % (v["id"], v["kernel"], v["kernel_ms"], v["all_kernels_ms"], v["step_ms_host_clock"], v["frac"], v["frac_dominant_kernel"] and round(v["frac_dominant_kernel"], 3), v["bit_exact_vs_oracle"]))
PY
  ;;
prof)
  echo "== rocprofv3 --kernel-trace --stats of the driver command line =="; cd /tmp && export TMPDIR=/tmp
  rm -rf $OUT/prof; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ${PROF_ARGS:-} > $OUT/prof_bench.json 2> $OUT/prof.err
  tail -2 $OUT/prof.err; find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12; cd $GRAFT_REPO_ROOT ;;
*) echo "unknown step $step" ;;
esac
done
