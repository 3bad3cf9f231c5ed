#!/bin/bash
# One bench line per workload (GPU box):  tools/bench_all.sh out.jsonl   -- kernel numbers only (--no-cpu)
set -u
OUT=${1:-gpurun_out/bench_all.jsonl}
: > $OUT
for w in cfg1_tsukuba_size_w7_d30 cfg2c_1080p_w7_d30 cfg3c_4k_w7_d32 cfg2_1080p_w9_d64 cfg2b_1080p_w9_d128 cfg4_8k_w21_d256 cfg5_kitti_w7_d128 cfg3_4k_w15_d128; do
  timeout -k 10 150 python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu 2> gpurun_out/bench_all.err | grep '^{' >> $OUT || { echo "$w failed"; tail -5 gpurun_out/bench_all.err; exit 1; }
  echo "$w done"
done
