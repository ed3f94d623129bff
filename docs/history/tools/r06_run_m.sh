#!/bin/bash
# On the GPU box (round 6, call m): the data-parallel tests and the DP program's evidence with the critic's lane gated from 1024 frames on.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_bench_multirank.py tests/test_gpu_fullsize.py -q 2>&1 | tail -6 > gpurun_out/gpu_dp_r06m.txt
cat gpurun_out/gpu_dp_r06m.txt
timeout 900 bash tools/dp_stats_ab.sh > gpurun_out/dp_plan_r06m.txt 2>&1
grep -v "^$" gpurun_out/dp_plan_r06m.txt | head -14
