#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for pr in 2 4 8 16 32; do
  for d in -1 0; do
    r=$(SADBM_BENCH_ALLOW_NO_TRAFFIC=1 SADBM_DIRECT=$d timeout -k 10 150 python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu --workload cfg3_4k_w15_d128 --pairs $pr 2>/dev/null) || { echo "pairs=$pr direct=$d FAILED"; continue; }
    echo "pairs=$pr direct=$d $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "Gd/s step_ms", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "prep_ms", d["roofline"]["prep_ms"], d["config"]["kernel"])')"
  done
done
