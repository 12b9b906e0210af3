#!/bin/bash
# First GPU session: sanity -> microbench -> parity tests -> bench -> rocprof.  Everything under timeouts.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
echo "== host =="; nproc; free -g | head -2; rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== microbench =="
timeout 300 ./tools/microbench > $OUT/microbench.jsonl 2>&1; tail -40 $OUT/microbench.jsonl
echo "== pytest gpu =="
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
echo "== bench 100M =="
timeout 600 python bench.py --rows 100000000 --steps 10 --warmup 2 --extra > $OUT/bench_100M.json 2> $OUT/bench_100M.err; cat $OUT/bench_100M.json; tail -12 $OUT/bench_100M.err
echo "== bench 1B =="
timeout 900 python bench.py --steps 20 --warmup 3 --extra > $OUT/bench_1B.json 2> $OUT/bench_1B.err; cat $OUT/bench_1B.json; tail -12 $OUT/bench_1B.err
echo "== bench 1B no-dma =="
PINOT_GPU_NO_DMA=1 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_1B_nodma.json 2> $OUT/bench_1B_nodma.err; cat $OUT/bench_1B_nodma.json; tail -3 $OUT/bench_1B_nodma.err
echo "== rocprof =="
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_r1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/rocprof.log 2>&1
tail -3 $OUT/rocprof.log
find $OUT/prof_r1 -name "*stats*" | head; for f in $(find $OUT/prof_r1 -name "*kernel_stats*.csv"); do head -12 $f; done
find $OUT/prof_r1 -name "*kernel_trace*.csv" -size +20M -delete
