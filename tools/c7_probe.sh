#!/bin/bash
# tools/c7_probe.sh <tag>: typed / plane tests, then the C7 configurations (LONG / DOUBLE metric columns, 250 M rows)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_typed.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python tools/bench_configs.py --match "C7" --only c7 --out gpurun_out/$1/c7.jsonl 2> gpurun_out/$1/c7.err | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    if 'config' in d: print('   %-74s k=%.3f GBps=%.0f frac=%.3f exact=%s %s' % (d['config'][:74], d['kernel_ms'], d.get('GBps', 0), d.get('GBps', 0) / 8000.0, d.get('bit_exact_vs_oracle'), d.get('kernel', '')))"
tail -3 gpurun_out/$1/c7.err
