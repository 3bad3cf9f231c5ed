#!/bin/bash
# rows per block (SADBM_STRIP_TY) swept on batches of cfg1-size frames (384x288, 7x7, D=30): kernel ms per launch
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for pr in ${PAIRS:-256 512}; do
  for ty in ${TYS:-0 10 12 14 16 18 20 22 24 26 29 32 36 41 47 57 71 94}; do
    r=$(SADBM_BENCH_ALLOW_NO_TRAFFIC=1 SADBM_STRIP_TY=$ty timeout -k 10 100 python3 $ROOT/bench.py --no-cpu --steps 20 --warmup 3 --workload cfg1_tsukuba_size_w7_d30 --pairs $pr 2>/dev/null) || { echo "pairs=$pr ty=$ty FAILED"; continue; }
    echo "pairs=$pr ty=$ty $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["roofline"]["frac"])')"
  done
done
