#!/bin/bash
# One parameterised GPU job script (replaces the one-off tools/gpu_job<N>.sh files of rounds 1-4).
# Usage on the GPU box, from the repo root:   bash tools/gpu_job.sh <tag> <step> [<step> ...]
# Every step writes under gpurun_out/<tag>/ and prints a short tail; steps run in the order given.
#   gp_tol        GP PCG-tolerance sweep against the cached oracle results (tools/exp_gp_gpu_tolerance.py)
#   chain[:args]  chained RA->GP->BA against the frozen oracle chain(s) (tools/exp_chain_gpu.py)
#   tests[:expr]  pytest -m gpu (optionally -k expr)
#   testfiles:a.py:b.py   pytest -m gpu on those files, without -x
#   testsk:expr   pytest -m gpu -k expr, without -x
#   bench         python bench.py (default line)
#   bench_fast    python bench.py --no-extra --no-cpu-baseline
#   multirank:2:4 bench.py --gpus N, N ranks sharing this device (peer transport)
#   profile       tools/profile_all.sh <tag> (kernel trace + the two PMC passes of the headline workload)
#   py:<script>   python <script> (arguments after further colons)
set -u
TAG=$1
shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for step in "$@"; do
  name=${step%%:*}
  arg=""
  [ "$step" != "$name" ] && arg=${step#*:}
  echo "=== $step"
  case $name in
    gp_tol)
      timeout 900 python tools/exp_gp_gpu_tolerance.py $arg > $OUT/gp_tol.log 2>&1; tail -40 $OUT/gp_tol.log ;;
    chain)
      for g in tests/golden/chain_2k_oracle.npz:2000:200000 tests/golden/chain_c4_oracle.npz:10000:1000000; do
        IFS=: read f n p <<< "$g"
        [ -f $f ] && timeout 900 python tools/exp_chain_gpu.py $f $n $p $arg >> $OUT/chain.log 2>&1
      done
      tail -20 $OUT/chain.log ;;
    tests)
      if [ -n "$arg" ]; then timeout 2400 python -m pytest tests -m gpu -x -q -s -k "$arg" > $OUT/tests.log 2>&1
      else timeout 2400 python -m pytest tests -m gpu -x -q -s --durations=14 > $OUT/tests.log 2>&1; fi
      grep "\[parity\]" $OUT/tests.log > $OUT/tests_parity.txt; tail -15 $OUT/tests.log ;;
    testsk)      # pytest -m gpu -k <expr>, every failure reported (no -x)
      timeout 2400 python -m pytest tests -m gpu -q -s -k "$arg" > $OUT/testsk.log 2>&1
      grep "\[parity\]" $OUT/testsk.log > $OUT/testsk_parity.txt; grep -E "^(FAILED|ERROR)" $OUT/testsk.log; tail -3 $OUT/testsk.log ;;
    testfiles)   # pytest -m gpu on the given files (colon-separated), every failure reported (no -x)
      timeout 2400 python -m pytest ${arg//:/ } -m gpu -q -s > $OUT/testfiles.log 2>&1
      grep "\[parity\]" $OUT/testfiles.log > $OUT/testfiles_parity.txt; grep -E "^(FAILED|ERROR)" $OUT/testfiles.log; tail -5 $OUT/testfiles.log ;;
    bench)
      timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-600 $OUT/bench.json; tail -3 $OUT/bench.err ;;
    bench_fast)
      timeout 600 python bench.py --no-extra --no-cpu-baseline $arg > $OUT/bench_fast.json 2> $OUT/bench_fast.err; cut -c1-1500 $OUT/bench_fast.json; tail -3 $OUT/bench_fast.err ;;
    multirank)   # bench.py --gpus N with all N ranks on this one device (control flow + numerics of the N > 1 path, not speed)
      for n in ${arg//:/ }; do
        GSFM_BENCH_SINGLE_DEVICE=1 GSFM_BENCH_TRANSPORT=peer timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
          --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 2 --warmup 1 --no-extra --no-cpu-baseline \
          > $OUT/multirank_$n.log 2>&1
        grep "\"metric\"" $OUT/multirank_$n.log >> $OUT/bench_gpus_N_on_one_device.jsonl; tail -2 $OUT/multirank_$n.log | cut -c1-400
      done ;;
    profile)
      bash tools/profile_all.sh $TAG/prof pipeline_c4 > $OUT/profile.log 2>&1; tail -5 $OUT/profile.log ;;
    py)
      timeout 1200 python ${arg//:/ } > $OUT/py_$(basename ${arg%%:*} .py).log 2>&1; tail -30 $OUT/py_$(basename ${arg%%:*} .py).log ;;
    *) echo "unknown step $name" ;;
  esac
done
