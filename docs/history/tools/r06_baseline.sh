#!/bin/bash
# On the GPU box: GPU suite + kernel-by-kernel timelines (32 / 512 / 4096 frames) + minibatch sweep of the tree as it is.  usage: bash tools/r06_baseline.sh <tag>
TAG=${1:-r06a}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/gpu_suite_$TAG.txt
cat gpurun_out/gpu_suite_$TAG.txt
timeout 900 bash tools/prof_timelines.sh $TAG
