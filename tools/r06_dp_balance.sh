#!/bin/bash
# On the GPU box: the data-parallel program (one-rank RCCL group) with the balanced node numbering on / off at shard sizes, alternating.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/dp_balance.txt
: > $OUT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %5s : %8.2f steps/s  %.4f ms/step' % ('$1', '$2', d['value'], d['ms_per_step']))"; }
for r in 1 2; do
  for mb in 4096 2048 1024 512; do
    GRL_BALANCE_NODE_ORDER=0 python bench.py --dp-plan --minibatch $mb --steps 60 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | line natural $mb >> $OUT
    python bench.py --dp-plan --minibatch $mb --steps 60 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | line balanced $mb >> $OUT
    GRL_DP_GATE_FROM=1 python bench.py --dp-plan --minibatch $mb --steps 60 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | line bal+gate $mb >> $OUT
  done
done
cat $OUT
