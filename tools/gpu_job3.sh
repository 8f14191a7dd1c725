#!/bin/bash
# round-4 GPU job 3: convergence test moved into the vector kernels' last block; null arrays; knobs; peer fixes
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job3
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gp_gpu.py tests/test_ba_gpu.py tests/test_ra_gpu.py tests/test_edge_cases_gpu.py tests/test_multirank_gpu.py tests/test_adapter.py tests/test_rigs.py tests/test_ra_rigs.py -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -6 $OUT/tests.log
timeout 300 python bench.py --steps 5 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python tools/bench_kernels_summary.py $OUT/bench.json
