#!/bin/bash
# round-4 GPU job 5: whole -m gpu suite on the work in progress (closed-form A W for shared intrinsics blocks, parked-sum
# combine that keeps k_ba_build_cam at two waves per SIMD) + one bench line
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job5
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -6 $OUT/tests.log
timeout 300 python bench.py --steps 5 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python tools/bench_kernels_summary.py $OUT/bench.json
