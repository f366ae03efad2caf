#!/bin/bash
# Kernel trace of a short default bench run + per-kernel breakdown of one training step.
# usage (on the GPU box, from the repo root): bash tools/prof_step.sh <tag> [extra bench.py flags]
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o run -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --rl-envs 0 --a3c-envs 0 --settle-seconds 0 --no-kernel-timing --no-extra-runs "$@" > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
python3 $R/profiles/summarize_trace.py $R/gpurun_out/prof_$TAG/run_kernel_trace.csv > $R/gpurun_out/${TAG}_step_breakdown.txt
cp $R/gpurun_out/prof_$TAG/run_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
rm -f $R/gpurun_out/prof_$TAG/run_kernel_trace.csv
head -14 $R/gpurun_out/${TAG}_step_breakdown.txt
