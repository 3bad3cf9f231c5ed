#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  tools/prof.sh <tag> [bench args...]
# 1) kernel trace + stats   2) PMC passes (each its own run, counters within the per-block limits)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--no-cpu --steps 5 --warmup 2 $*"
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
         "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -- python3 $ROOT/bench.py $ARGS > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.log; exit 1; }
done
python3 $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
