#!/bin/bash
# tools/ia_regs.sh <unit> <kernel regexp> [-D...]: VGPRs / scratch / spills of the kernels of ONE unit whose name matches (compiler remarks, no GPU)
cd "$(dirname "$0")/../pinot_amd/csrc"
u=$1; k=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only "$@" -Rpass-analysis=kernel-resource-usage -S -o /tmp/${u}.s $u.hip 2>&1 | \
  awk -v k="$k" '/Function Name:/ {on = ($0 ~ k); if (on) {split($0, a, "Function Name: "); printf "%s", substr(a[2], 1, 90)}} on && /(VGPRs|ScratchSize|Occupancy|VGPRs Spill|SGPRs Spill|LDS Size)[^:]*: / {match($0, /remark: +[A-Za-z ]+(\[[a-z\/A-Z]+\])?: [0-9]+/); s = substr($0, RSTART + 8, RLENGTH - 8); gsub(/ +/, " ", s); printf " |%s", s} on && /LDS Size/ {print ""}'
