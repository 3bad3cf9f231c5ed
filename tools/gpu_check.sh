#!/bin/bash
# Round check on the GPU box: GPU parity tests, one bench line per workload, then the profile recipe.
#   tools/gpu_check.sh <tag>
set -u
TAG=${1:-check}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/tests_$TAG.log 2>&1 || { echo "tests failed"; tail -20 $OUT/tests_$TAG.log; exit 1; }
tail -2 $OUT/tests_$TAG.log
: > $OUT/bench_$TAG.jsonl
for wl in cfg3_4k_w15_d128 cfg2_1080p_w9_d64 cfg2b_1080p_w9_d128 cfg5_kitti_w7_d128 cfg4_8k_w21_d256; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu --workload $wl >> $OUT/bench_$TAG.jsonl 2> $OUT/bench_$TAG.err || { echo "bench $wl failed"; tail -5 $OUT/bench_$TAG.err; exit 1; }
done
python3 - <<PY
import json
for l in open("$OUT/bench_$TAG.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"].split(":")[0], d["value"], "Gd/s", d["frames_per_s"], "fps kernel_ms", d["roofline"]["kernel_ms"], "prep_ms", d["roofline"]["prep_ms"], d["config"]["kernel"], d["verified_bit_exact"])
PY
