#!/bin/bash
# Kernel stats of the DEFAULT bench command (30 steps, 5 warm-up; GPU box):  tools/prof_headline.sh <tag>
#   rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu ; prints the stats CSV rows and the live HIP-event time of the same run
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/headline_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu > $OUT/bench.log 2>&1 || { echo "failed"; tail -5 $OUT/bench.log; exit 1; }
{ echo "== rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu   (default: --steps 30 --warmup 5, cfg3_4k_w15_d128, 32 pairs per step)";
  cat $(find $OUT/trace -name '*kernel_stats.csv' | head -1);
  echo "bench.py live HIP-event timing in the same run: $(grep -o '"kernel_ms": [0-9.]*' $OUT/bench.log)  $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench.log)  $(grep -o '"value": [0-9.]*' $OUT/bench.log | head -1)"; } > $OUT/summary.txt
cat $OUT/summary.txt
