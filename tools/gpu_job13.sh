#!/bin/bash
# round-4 GPU job 13: k_gp_build_cam carrying the camera half of the linearisation and the closed-form gauge products
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job13
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gp_gpu.py tests/test_edge_cases_gpu.py tests/test_multirank_gpu.py tests/test_golden.py tests/test_pipeline_gpu.py tests/test_rigs.py tests/test_scene_level_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "gp_" -s > $OUT/tests_full_gp.log 2>&1
echo "tests rc=$?" >> $OUT/tests_full_gp.log
grep "parity\|passed\|failed\|rc=" $OUT/tests_full_gp.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o gp -- python $GRAFT_REPO_ROOT/tools/ab_gp_sweeps.py 0 > $OUT/gp_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $OUT/gp_results.db > $OUT/gp_kernel_stats.csv
rm -f $OUT/*.db
head -14 $OUT/gp_kernel_stats.csv | cut -c1-60,100-200
timeout 300 python tools/ab_gp_sweeps.py 0 | tail -1
