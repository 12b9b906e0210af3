#!/bin/bash
# FETCH_SIZE / TCC request counters of the sparse-read patterns of tools/microbench.hip (mode "sparse"): what the counter reports for a
# known number of touched 128-byte lines.  Leaves gpurun_out/<round>/fetch_calibration.json (copy into profiles/<round>/).
R=${1:-r4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for counters in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_MISS_sum TCC_HIT_sum"; do
  tag=$(echo $counters | tr ' ' '_')
  rm -rf $OUT/cal_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $OUT/cal_$tag -o cal -- $GRAFT_REPO_ROOT/tools/microbench sparse > $OUT/cal_$tag.log 2>&1
  tail -1 $OUT/cal_$tag.log
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections, re
out = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/cal_*/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "sparse_read" not in k and "stream_read" not in k:
            continue
        m = re.search(r"sparse_read<(\d+), (\d+)>", r["Kernel_Name"])
        name = ("sparse_%sB_every_%sB" % (m.group(2), m.group(1))) if m else "stream_read_16B_per_lane"
        out[name][r["Counter_Name"]] = out[name].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
json.dump(out, open(sys.argv[1] + "/fetch_calibration.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
