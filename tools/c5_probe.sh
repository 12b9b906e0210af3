#!/bin/bash
# tools/c5_probe.sh <tag>: index-AND tests, then the C5 variants of bench.py's line (parity on), summarised
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_index_and.py tests/test_gpu_parity.py -m gpu -x -q -k "index or inverted or and_of" 2>&1 | tail -6
timeout 900 python bench.py --steps 5 --warmup 2 --variants "${2:-C5}" > gpurun_out/$1/bench.json 2> gpurun_out/$1/bench.err || tail -5 gpurun_out/$1/bench.err
python - gpurun_out/$1/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
for v in d.get("variants", []):
    print("%-16s %-28s k=%.4f all=%.4f frac=%.3f exact=%s matched=%d" % (v["id"], v["kernel"], v["kernel_ms"], v["all_kernels_ms"], v["frac"], v["bit_exact_vs_oracle"], v["docs_matched"]))
PY
