#!/bin/bash
# round-4 GPU job 17: whole -m gpu suite + smoke
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job17
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -x -q -m gpu -s > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log
grep "passed\|failed\|rc=" $OUT/tests.log | tail -3
grep "\[parity\]" $OUT/tests.log > $OUT/parity.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/smoke.log
tail -3 $OUT/smoke.log
