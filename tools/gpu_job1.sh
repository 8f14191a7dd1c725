#!/bin/bash
# round-4 GPU job 1: the N > 1 paths on one device + the gather calibration
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export GSFM_BENCH_SINGLE_DEVICE=1
timeout 120 tools/exp_gather_calib 10 > $OUT/gather_calib_times.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o calib_FETCH_SIZE -- $GRAFT_REPO_ROOT/tools/exp_gather_calib 2 > $OUT/calib_f.log 2>&1;
 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o calib_WRITE_SIZE -- $GRAFT_REPO_ROOT/tools/exp_gather_calib 2 > $OUT/calib_w.log 2>&1)
python tools/gather_calib_summary.py $OUT/calib_FETCH_SIZE_results.db $OUT/calib_WRITE_SIZE_results.db > $OUT/gather_calib_pmc.csv 2> $OUT/gather_calib_pmc.err
rm -f $OUT/*.db
timeout 900 python -m pytest tests/test_multirank_gpu.py -x -q -k "above_the_single" > $OUT/multirank_big.log 2>&1
echo "multirank rc=$?" >> $OUT/multirank_big.log
for n in 2 4 8; do
  GSFM_BENCH_TRANSPORT=peer timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 2 --warmup 1 > $OUT/bench_peer_n$n.json 2> $OUT/bench_peer_n$n.err
  echo "n=$n rc=$?" >> $OUT/bench_rc.txt
done
GSFM_BENCH_TRANSPORT=rccl timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29510 bench.py --gpus 2 --steps 1 --warmup 1 > $OUT/bench_rccl_n2.json 2> $OUT/bench_rccl_n2.err
echo "rccl n=2 rc=$?" >> $OUT/bench_rc.txt
timeout 400 python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -3 $OUT/multirank_big.log; cat $OUT/bench_rc.txt; cat $OUT/gather_calib_times.txt; cat $OUT/gather_calib_pmc.csv
