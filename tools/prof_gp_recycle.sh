# Round 6: rocprofv3 kernel trace of configs[3] GP with the recycled-vector preconditioner on and off (gpurun_out/r06k/).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in on off; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o rc_$mode -- python $GRAFT_REPO_ROOT/tools/exp_gp_recycle_gpu.py --only $mode 10000 1000000 0 > $OUT/rc_$mode.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/rocpd_stats.py $OUT/rc_${mode}_results.db > $OUT/rc_${mode}_kernel_stats.csv
  cd /tmp
done
rm -f $OUT/*.db
ls $OUT
