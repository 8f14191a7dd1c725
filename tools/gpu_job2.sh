#!/bin/bash
# round-4 GPU job 2: XCD-affine point chunks of the camera-major order — correctness on small problems, A/B at configs[3]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job2
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
GSFM_CAM_CHUNKS=8 timeout 600 python -m pytest tests/test_gp_gpu.py tests/test_ba_gpu.py tests/test_edge_cases_gpu.py tests/test_rigs.py -x -q > $OUT/tests_chunks8.log 2>&1
echo "tests rc=$?" >> $OUT/tests_chunks8.log
tail -5 $OUT/tests_chunks8.log
for c in 1 16 32 64; do
  GSFM_CAM_CHUNKS=$c timeout 300 python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench_chunks$c.json 2> $OUT/bench_chunks$c.err
done
timeout 300 python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench_policy.json 2> $OUT/bench_policy.err
python tools/bench_kernels_summary.py $OUT/bench_chunks1.json $OUT/bench_chunks16.json $OUT/bench_chunks32.json $OUT/bench_chunks64.json $OUT/bench_policy.json
