#!/bin/bash
# A/B builds of libpinot_gpu.so on the same box: tools/gpu_ab.sh "<variant.so> ..." [rounds]   ("" = the in-tree default build)
cd $GRAFT_REPO_ROOT
R=${2:-2}
for i in $(seq 1 $R); do
  for lib in "" $1; do
    echo "== lib=${lib:-default} round $i"
    PINOT_GPU_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --extra 2> gpurun_out/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   headline kernel_ms=%.4f parity=%s' % (d['roofline']['kernel_ms'], d.get('parity',{}).get('bit_exact_vs_oracle')))"
    grep extra gpurun_out/ab.err | python -c "
import sys,json
print('   ' + '  '.join('%s %.3f' % (json.loads(l)['extra'][:22], json.loads(l)['kernel_ms']) for l in sys.stdin))"
  done
done
