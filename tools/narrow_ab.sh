#!/bin/bash
# tools/narrow_ab.sh <tag> [variant.so]: scan_narrow_kernel vs scan_private_kernel on 4 / 6 / 8-bit columns, same build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$1
for v in 1 0; do
  echo "== PINOT_GPU_SCAN_NARROW=$v"
  PINOT_GPU_SCAN_NARROW=$v timeout 600 python tools/narrow_probe.py > gpurun_out/$1/narrow_kernel$v.jsonl 2> gpurun_out/$1/narrow$v.err
  cut -c1-220 gpurun_out/$1/narrow_kernel$v.jsonl
done
echo "== PINOT_GPU_SCAN_NARROW_SINGLE=0 (single leaves in the general narrow kernel)"
PINOT_GPU_SCAN_NARROW_SINGLE=0 timeout 600 python tools/narrow_probe.py > gpurun_out/$1/narrow_kernel_single0.jsonl 2> gpurun_out/$1/narrow_single0.err
cut -c1-220 gpurun_out/$1/narrow_kernel_single0.jsonl
if [ -n "$2" ]; then
  echo "== variant $2"
  PINOT_GPU_LIB=$GRAFT_REPO_ROOT/$2 timeout 600 python tools/narrow_probe.py > gpurun_out/$1/narrow_kernel_variant.jsonl 2> gpurun_out/$1/narrow_variant.err
  cut -c1-220 gpurun_out/$1/narrow_kernel_variant.jsonl
fi
python - <<PY
import json
a = [json.loads(l) for l in open("gpurun_out/$1/narrow_kernel1.jsonl")]
b = [json.loads(l) for l in open("gpurun_out/$1/narrow_kernel0.jsonl")]
print("answers identical:", all(x["count"] == y["count"] for x, y in zip(a, b)) and len(a) == len(b) > 0)
PY
