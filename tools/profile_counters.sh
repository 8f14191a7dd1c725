#!/bin/bash
# Stall / cache counters of the sweep kernels (next step for k_gp_phaseA): one rocprofv3 --pmc pass per
# counter group (kernel-trace only, as the pool requires), summed per kernel by tools/pmc_counters.py.
# Usage (on the GPU box, from the repo root): bash tools/profile_counters.sh <tag> [workload]   (default gp_c3)
#   SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY  — WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES
#   SQ_INSTS_VALU, SQ_INSTS_VMEM_RD, SQ_INSTS_LDS, SQ_INSTS_SALU        — instruction mix
#   TA_BUSY_sum, TCP_TCC_READ_REQ_sum, TCC_HIT_sum, TCC_MISS_sum          — gather path and L2 hit rate
set -u
TAG=${1:-counters}
WL=${2:-gp_c3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
           "TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $OUT -o ${WL}_$name -- \
    python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 1 --warmup 0 --no-extra --no-cpu-baseline > $OUT/${WL}_$name.log 2>&1
done
cd $GRAFT_REPO_ROOT
mkdir -p $OUT/summary
python tools/pmc_counters.py $OUT > $OUT/summary/${WL}_counters.csv 2> $OUT/summary/${WL}_counters.err
rm -f $OUT/*.db
ls $OUT/summary
