#!/bin/bash
# round 5, first GPU call: GPU suite on the pruned build, the new bench line, PMC tables (with pass E) for the headline and for config 5's bf16 build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | tr '\n' ' ') > gpurun_out/sq_counters.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite_a.txt 2>&1; echo "suite rc $?" >> gpurun_out/gpu_suite_a.txt
tail -5 gpurun_out/gpu_suite_a.txt
timeout 900 bash tools/profile_round.sh r05a
timeout 600 bash tools/profile_workload.sh rope_hepi_bf16
python bench.py --workload rope_hepi_bf16 > gpurun_out/bench_line_rope_hepi_bf16_r05a.json 2> gpurun_out/bench_rope_bf16_r05a.err
tail -c 600 gpurun_out/bench_line_rope_hepi_bf16_r05a.json
