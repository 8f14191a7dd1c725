#!/bin/bash
# round-4 GPU job 11: draw orders + adapter numbering; shared-intrinsics BA against the regenerated fixtures; setup / solve split
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job11
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gp_gpu.py tests/test_adapter.py tests/test_flatio.py tests/test_scene_level_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "shared" -s > $OUT/tests_full_shared.log 2>&1
echo "tests rc=$?" >> $OUT/tests_full_shared.log
grep "parity\|passed\|failed\|rc=\|assert\|Error" $OUT/tests_full_shared.log | head -20
timeout 300 python tools/exp_setup_time.py > $OUT/setup_time.txt 2>&1
cat $OUT/setup_time.txt
