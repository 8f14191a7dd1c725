#!/bin/bash
# round-4 GPU job 20: evidence of the round — kernel stats + PMC passes of the headline command, then the default bench line
set -u
cd $GRAFT_REPO_ROOT
bash tools/profile_all.sh r04_final pipeline_c4 > gpurun_out/r04_final_profile.log 2>&1
tail -3 gpurun_out/r04_final_profile.log
mkdir -p gpurun_out/r04_final
timeout 900 python bench.py > gpurun_out/r04_final/bench_default.json 2> gpurun_out/r04_final/bench_default.err
echo "bench rc=$?"
python tools/bench_kernels_summary.py gpurun_out/r04_final/bench_default.json
