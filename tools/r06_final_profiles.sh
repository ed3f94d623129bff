#!/bin/bash
# round 6, evidence set of the final build on ONE box: GPU suite, headline line + rocprofv3 kernel stats + PMC passes (the summary records the
# library's source hash), the other workloads' lines, the minibatch sweep, timelines, the data-parallel program on a one-rank RCCL group.
cd $GRAFT_REPO_ROOT
V=${1:-r06v1}
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/gpu_suite_$V.txt 2>&1; echo "suite rc $?" >> gpurun_out/gpu_suite_$V.txt
tail -4 gpurun_out/gpu_suite_$V.txt
timeout 1200 bash tools/profile_round.sh $V
# config 5's bf16 build: kernel statistics are in its bench line; the PMC passes for its roofline.traffic
export GRL_WORKLOAD=rope_hepi_bf16
rm -rf gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC gpurun_out/pmcD gpurun_out/pmcE
timeout 900 bash tools/pmc_passes.sh
python tools/pmc_report.py gpurun_out gpurun_out/pmc_summary_rope_hepi_bf16_$V.json > gpurun_out/pmc_table_rope_hepi_bf16_$V.txt
rm -rf gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC gpurun_out/pmcD gpurun_out/pmcE
unset GRL_WORKLOAD
timeout 1500 bash tools/bench_all.sh $V
GRL_TL_SIZES="32 512 4096" timeout 900 bash tools/prof_timelines.sh $V
timeout 900 bash tools/dp_stats_ab.sh > gpurun_out/dp_plan_$V.txt 2>&1
tail -30 gpurun_out/dp_plan_$V.txt
