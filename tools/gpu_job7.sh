#!/bin/bash
# round-4 GPU job 7: GP phase B in the chunked (XCD-partitioned) order
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job7
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
GSFM_KNOBS=chunked_sweeps=1 timeout 900 python -m pytest tests/test_gp_gpu.py tests/test_edge_cases_gpu.py tests/test_multirank_gpu.py tests/test_golden.py tests/test_pipeline_gpu.py -x -q -m gpu > $OUT/tests_forced.log 2>&1
echo "tests rc=$?" >> $OUT/tests_forced.log
tail -4 $OUT/tests_forced.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "gp_" -s > $OUT/tests_full_gp.log 2>&1
echo "tests rc=$?" >> $OUT/tests_full_gp.log
grep "parity\|passed\|failed\|rc=" $OUT/tests_full_gp.log
timeout 300 python bench.py --steps 5 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python tools/bench_kernels_summary.py $OUT/bench.json
GSFM_KNOBS=chunked_sweeps=2 timeout 300 python bench.py --steps 5 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench_off.json 2> $OUT/bench_off.err
python tools/bench_kernels_summary.py $OUT/bench_off.json
