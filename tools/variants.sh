#!/bin/bash
# Runs bench.py once per experiment build (opencv_amd/csrc/libsadbm*.so); prints value + kernel ms.
#   tools/variants.sh [workload] [flags]      (experiment builds may be deliberately wrong: )
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-cfg3_4k_w15_d128}
FL=${2:-0}
for so in $ROOT/opencv_amd/csrc/libsadbm*.so; do
  r=$(SADBM_LIB=$so timeout -k 10 120 python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu --workload $WL --flags $FL 2>/dev/null) || { echo "$(basename $so) FAILED"; continue; }
  echo "$(basename $so) $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"], d["config"]["kernel"])')"
done
