#!/bin/bash
# round-4 GPU job 9: kernel-trace of GP alone at configs[3] size (where do the 223 ms of a solve go?)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job9
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o gp -- python $GRAFT_REPO_ROOT/tools/ab_gp_sweeps.py 0 > $OUT/gp_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $OUT/gp_results.db > $OUT/gp_kernel_stats.csv
rm -f $OUT/*.db
head -40 $OUT/gp_kernel_stats.csv
tail -2 $OUT/gp_trace.log
