#!/bin/bash
# tools/pmc_kernel.sh <out-dir> <kernel-substring> <bench --variants regexp> <counter set> [<counter set> ...]
# One rocprofv3 --pmc pass per counter set over `bench.py --variants <regexp>`; prints the per-dispatch mean of every counter over the
# dispatches whose kernel name contains the substring.  (--pmc with --kernel-trace only: MI355X_MICROARCH.md's recipe.)
set -u
out=$1; kern=$2; variants=$3; shift 3
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf $out/pmc_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --variants "$variants" --segments 1 --steps 3 --warmup 1 --no-cpu-baseline --no-clock-settle > $out/pmc_$i.log 2>&1
  for f in $(find $out/pmc_$i -name "*counter_collection*.csv"); do python3 - "$f" "$kern" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if sys.argv[2] in r['Kernel_Name']:
        agg[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
for c, d in sorted(agg.items()):
    vals = list(d.values())
    print('   %-28s per-dispatch mean %.5g (n=%d)' % (c, sum(vals) / len(vals), len(vals)))
PY
  done
done
