#!/bin/bash
# One gpurun call: kernel stats (rocprofv3 --kernel-trace --stats) and the PMC passes of a few steps of ONE bench workload.
#   bash tools/profile_workload.sh rope_hepi_bf16      -> gpurun_out/<wl>_kernel_stats.csv, gpurun_out/pmc_<wl>_table.txt, pmc_<wl>_summary.json
WL=${1:-rope_hepi_bf16}
R=$GRAFT_REPO_ROOT
export GRL_WORKLOAD=$WL GRL_STEPS=3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$WL -o st -- python3 $R/tools/profile_step.py > $R/gpurun_out/stats_$WL.log 2>&1
cd $R
cp $(find gpurun_out/stats_$WL -name '*kernel_stats.csv' | head -1) gpurun_out/${WL}_kernel_stats.csv
GRL_STEPS=2 bash tools/pmc_cmd.sh $WL tools/profile_step.py | grep -E "edge|node_mlp|fiber|lift|kernel \|" | cut -c1-240
head -12 gpurun_out/${WL}_kernel_stats.csv | cut -c1-160
