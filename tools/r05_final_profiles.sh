#!/bin/bash
# round 5, evidence set of the final build on ONE box: GPU suite, headline line + rocprofv3 kernel stats + PMC passes, the other workloads'
# lines, sizes, timelines, the data-parallel program on a one-rank RCCL group
cd $GRAFT_REPO_ROOT
V=${1:-r05c}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite_$V.txt 2>&1; echo "suite rc $?" >> gpurun_out/gpu_suite_$V.txt
tail -3 gpurun_out/gpu_suite_$V.txt
timeout 900 bash tools/profile_round.sh $V
timeout 1500 bash tools/bench_all.sh $V
GRL_TL_SIZES="512 4096" timeout 600 bash tools/prof_timelines.sh $V
timeout 600 bash tools/dp_stats_ab.sh > gpurun_out/dp_plan_$V.txt 2>&1
tail -30 gpurun_out/dp_plan_$V.txt
