#!/bin/bash
# One gpurun call: bench line, rocprofv3 kernel stats of the same bench command, and the PMC passes.  Usage: bash tools/profile_round.sh v10
V=${1:-vX}
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/bench_line_$V.json 2> gpurun_out/bench_$V.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$V -o bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_prof_$V.log 2>&1
cd $R
bash tools/pmc_passes.sh
python tools/pmc_report.py gpurun_out gpurun_out/pmc_summary_$V.json > gpurun_out/pmc_table_$V.txt
tail -c 400 gpurun_out/bench_line_$V.json
