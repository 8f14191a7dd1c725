#!/bin/bash
# round-4 GPU job 22: the N > 1 path of the FINAL code on one device (2 and 4 ranks, peer mailboxes, acceptance step on)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_job22
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export GSFM_BENCH_SINGLE_DEVICE=1
for n in 2 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) bench.py --gpus $n --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench_peer_n$n.json 2> $OUT/bench_peer_n$n.err
  echo "n=$n rc=$?"
  python tools/bench_kernels_summary.py $OUT/bench_peer_n$n.json
done
