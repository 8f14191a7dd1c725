#!/bin/bash
# round-4 GPU job 28: L2 hit rates of the two GP sweeps at configs[3] size (final code) — one PMC pass, kernel-trace only
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job28
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace -d $OUT -o gp_c4_TCC -- python $GRAFT_REPO_ROOT/tools/ab_gp_sweeps.py 0 > $OUT/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_counters.py $OUT > $OUT/gp_c4_l2.csv 2> $OUT/gp_c4_l2.err
rm -f $OUT/*.db
head -6 $OUT/gp_c4_l2.csv
