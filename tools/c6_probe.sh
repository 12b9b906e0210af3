#!/bin/bash
# tools/c6_probe.sh <tag>: wide group-by tests, then the C6 configurations with the packed records on and off
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_group_map.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4
for v in 1; do
  echo "== PINOT_GPU_PARTITION_PACKED=$v"
  PINOT_GPU_PARTITION_PACKED=$v timeout 900 python tools/bench_configs.py --match "C6" --only c23 --out gpurun_out/$1/c6_packed$v.jsonl 2> gpurun_out/$1/c6_packed$v.err | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    if 'config' in d: print('   %-74s k=%.3f all=%.3f step=%.2f exact=%s' % (d['config'][:74], d['kernel_ms'], d.get('all_kernels_ms', 0), d.get('step_ms_host_clock', 0), d.get('bit_exact_vs_oracle')))"
done
