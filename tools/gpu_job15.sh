#!/bin/bash
# round-4 GPU job 15: where is the device idle inside a step? (kernel-trace timestamps of one pipeline run)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job15
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT -o pipe -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_gaps.py $OUT/pipe_results.db | tee $OUT/gaps.txt
rm -f $OUT/*.db
