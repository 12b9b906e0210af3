#!/bin/bash
# tools/c7_ab.sh <tag>: raw 8-byte columns, coalesced vs lane-contiguous reads (C7), after the typed tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_typed.py tests/test_gpu_nulls.py -m gpu -x -q 2>&1 | tail -3
for v in 1 0; do
  echo "== PINOT_GPU_RAW64_COALESCED=$v"
  PINOT_GPU_RAW64_COALESCED=$v timeout 900 python tools/bench_configs.py --match "C7.*(raw|no filter)" --only c7 --out gpurun_out/$1/c7_coalesced$v.jsonl 2> gpurun_out/$1/c7.err | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    if 'config' in d: print('   %-74s k=%.3f GBps=%.0f frac=%.3f exact=%s' % (d['config'][:74], d['kernel_ms'], d.get('GBps', 0), d.get('GBps', 0) / 8000.0, d.get('bit_exact_vs_oracle')))"
done
