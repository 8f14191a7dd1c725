#!/bin/bash
# round-4 GPU job 18: the point half of the linearisation rides on k_gp_build_track
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job18
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gp_gpu.py tests/test_edge_cases_gpu.py tests/test_multirank_gpu.py tests/test_golden.py tests/test_pipeline_gpu.py tests/test_rigs.py tests/test_scene_level_gpu.py tests/test_adapter.py -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
grep "passed\|failed\|rc=" $OUT/tests.log | tail -3
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "gp_" -s > $OUT/tests_full_gp.log 2>&1
echo "tests rc=$?" >> $OUT/tests_full_gp.log
grep "parity\|passed\|failed\|rc=" $OUT/tests_full_gp.log
timeout 300 python tools/ab_gp_sweeps.py 0 | tail -1
timeout 300 python bench.py --steps 5 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python tools/bench_kernels_summary.py $OUT/bench.json
