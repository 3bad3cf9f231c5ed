#!/bin/bash
# A/B of the staged kernel reading the caller's images in place (default) against the prep pass + padded planes (SADBM_DIRECT=0).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for wl in ${WLS:-cfg3_4k_w15_d128 cfg5_kitti_w7_d128 cfg4_8k_w21_d256 cfg2b_1080p_w9_d128}; do
  for d in 1 0; do
    r=$(SADBM_BENCH_ALLOW_NO_TRAFFIC=1 SADBM_DIRECT=$([ $d = 1 ] && echo -1 || echo 0) timeout -k 10 150 python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu --workload $wl 2>/dev/null) || { echo "$wl direct=$d FAILED"; continue; }
    echo "$wl direct=$d $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "Gd/s step_ms", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "prep_ms", d["roofline"]["prep_ms"], d["config"]["kernel"], d["verified_bit_exact"])')"
  done
done
