#!/bin/bash
# A/B two builds of libpinot_gpu.so on the same box: tools/gpu_ab.sh <variant.so> [rounds]
cd $GRAFT_REPO_ROOT
V=$1; R=${2:-2}
for i in $(seq 1 $R); do
  for lib in "" "$V"; do
    echo "== lib=${lib:-default} round $i"
    PINOT_GPU_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --extra 2> gpurun_out/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   headline kernel_ms=%.4f' % d['roofline']['kernel_ms'])"
    grep extra gpurun_out/ab.err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-34s %.3f ms' % (d['extra'], d['kernel_ms']))"
  done
done
echo "== C3 (default lib)"; timeout 600 python tools/bench_configs.py --match "C3|GROUP" --only c23 --no-check 2>&1 | grep -E "C3" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-52s %.3f ms %6.0f GB/s' % (d['config'], d['kernel_ms'], d['GBps']))"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_typed.py -m gpu -x -q -k "group" 2>&1 | tail -3
