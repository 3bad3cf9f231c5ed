#!/bin/bash
# Kernel stats of the secondary workloads (run on the GPU box):  tools/prof_stats.sh <tag>
#   one rocprofv3 --kernel-trace --stats run per workload; prints the stats CSV and the live HIP-event time of the same run
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stats_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
: > $OUT/summary.txt
for wl in cfg2_1080p_w9_d64 cfg2b_1080p_w9_d128 cfg4_8k_w21_d256 cfg5_kitti_w7_d128; do
  timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$wl -- python3 $ROOT/bench.py --no-cpu --steps 5 --warmup 2 --workload $wl > $OUT/$wl.log 2>&1 || { echo "$wl failed"; tail -5 $OUT/$wl.log; exit 1; }
  echo "== $wl  (rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --steps 5 --warmup 2 --workload $wl)" >> $OUT/summary.txt
  cat $(find $OUT/$wl -name '*kernel_stats.csv' | head -1) >> $OUT/summary.txt
  echo "bench.py live HIP-event timing in the same run: $(grep -o '"kernel_ms": [0-9.]*' $OUT/$wl.log)" >> $OUT/summary.txt
done
cat $OUT/summary.txt
