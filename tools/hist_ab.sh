#!/bin/bash
# A/B of scan_hist_kernel experiment builds (pinot_amd/csrc/build/x/libpinot_gpu_x*.so, -DPG_HIST_X=n) on the irregular-dictionary C2b.
cd $GRAFT_REPO_ROOT
for lib in "" $(ls pinot_amd/csrc/build/x/libpinot_gpu_x*.so 2>/dev/null); do
  for sel in ${SELS:-100}; do
    env PINOT_GPU_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 300 python bench.py --steps 20 --warmup 3 --dictionary ${DICT:-irregular} --threshold $sel --no-cpu-baseline > /tmp/ab.json 2> /tmp/ab.err
    python - "$lib" $sel <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/ab.json")); r = d["roofline"]
    print("%-50s sel=%s %s kernel_ms=%.4f frac=%.3f sum=%s" % (sys.argv[1] or "default", sys.argv[2], r["kernel"], r["kernel_ms"], r["frac"], d["result"]["sum"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/ab.err").read()[-400:])
PY
  done
done
