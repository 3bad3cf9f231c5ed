#!/bin/bash
# A/B of the LDS-staged matcher (flags 0) against the L1-operand variant (flags 4) on the bench workloads.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for wl in ${WLS:-cfg3_4k_w15_d128 cfg2_1080p_w9_d64 cfg5_kitti_w7_d128 cfg4_8k_w21_d256}; do
  for f in ${FLAGS:-0 4}; do
    r=$(timeout -k 10 120 python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu --workload $wl --flags $f 2>/dev/null) || { echo "$wl flags=$f FAILED"; continue; }
    echo "$wl flags=$f $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "Gd/s kernel_ms", d["roofline"]["kernel_ms"], "prep_ms", d["roofline"]["prep_ms"], d["verified_bit_exact"])')"
  done
done
