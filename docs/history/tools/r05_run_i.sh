#!/bin/bash
# full GPU suite + config 5 (bf16) evidence set: bench line, kernel stats, PMC table
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite_i.txt 2>&1; echo "suite rc $?" >> gpurun_out/gpu_suite_i.txt
tail -4 gpurun_out/gpu_suite_i.txt
python bench.py --workload rope_hepi_bf16 > gpurun_out/bench_line_rope_hepi_bf16_r05i.json 2> gpurun_out/bench_rope_bf16_r05i.err
tail -c 300 gpurun_out/bench_line_rope_hepi_bf16_r05i.json
timeout 600 bash tools/profile_workload.sh rope_hepi_bf16 | tail -12 | cut -c1-250
