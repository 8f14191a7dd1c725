#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/<tag>/: kernel-trace stats and the two PMC passes (separate runs,
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of the headline workload — the same command bench.py's default
# line comes from, minus the CPU baseline and the side measurements.
# Usage (on the GPU box, from the repo root): bash tools/profile_all.sh <tag> [workloads...]
set -u
TAG=${1:-prof}
shift
WLS=${@:-pipeline_c4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in $WLS; do
  steps=2; [ $wl = ra_c2 ] && steps=10
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o ${wl} -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps $steps --warmup 1 --no-extra --no-cpu-baseline > $OUT/${wl}_trace.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace -d $OUT -o ${wl}_$c -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 1 --warmup 0 --no-extra --no-cpu-baseline > $OUT/${wl}_$c.log 2>&1
  done
  tail -1 $OUT/${wl}_trace.log | cut -c1-200
done
ls $OUT | head -40
# summaries only travel back (the rocpd databases are tens of MB)
cd $GRAFT_REPO_ROOT
mkdir -p $OUT/summary
for wl in $WLS; do
  python tools/rocpd_stats.py $OUT/${wl}_results.db > $OUT/summary/${wl}_kernel_stats.csv
  python tools/pmc_traffic.py $OUT/${wl} > $OUT/summary/${wl}_pmc.csv 2> $OUT/summary/${wl}_pmc.err
  grep "\"metric\"" $OUT/${wl}_trace.log > $OUT/summary/${wl}_bench_line.json
done
cp profiles/pmc_traffic.json $OUT/summary/ 2>/dev/null
rm -f $OUT/*.db
