#!/bin/bash
# tools/prof_variants.sh <tag> <variants regexp>: rocprofv3 kernel stats of bench.py restricted to some variants
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o v -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --variants "$2" > $OUT/prof.log 2>&1
tail -2 $OUT/prof.log | cut -c1-200
find $OUT/prof -name "*kernel_stats*.csv" -exec cat {} \; | cut -c1-220 | head -${3:-16}
find $OUT/prof -name "*kernel_trace*.csv" -size +8M -delete
