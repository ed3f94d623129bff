#!/bin/bash
# On the GPU box: the data-parallel program's one-launch tail (Adam + reported values) replayed as a one-node graph or issued eagerly, alternating.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_dp_tail.txt
: > $OUT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %5s : %8.2f steps/s  %.4f ms/step' % ('$1', '$2', d['value'], d['ms_per_step']))"; }
for r in 1 2 3; do
  for mb in 512 1024; do
    GRL_DP_EAGER_TAIL=0 python bench.py --dp-plan --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | line graph $mb >> $OUT
    GRL_DP_EAGER_TAIL=1 python bench.py --dp-plan --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | line eager $mb >> $OUT
  done
done
cat $OUT
timeout 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_rollout.py -q 2>&1 | tail -4
GRL_DP_EAGER_TAIL=1 timeout 600 python -m pytest tests/test_gpu_dp.py -q 2>&1 | tail -3
