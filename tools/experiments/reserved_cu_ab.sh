#!/bin/bash
# On the GPU box: one compute unit left free by the ConvNeXt kernels (GRL_RESERVED_CUS=1) so that the critic lane's gate can wait inside a
# multi-step launch -- against the shipped forms (ungated multi-step launches below 3072 frames, the gated per-step program above).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_reserved_cu.txt
: > $OUT
GRL_RESERVED_CUS=1 timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_step.py tests/test_gpu_determinism.py -x -q 2>&1 | tail -4 | tee -a $OUT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-22s %5s : %8.2f steps/s  %.4f ms/step  %s' % ('$1', '$2', d['value'], d['ms_per_step'], d['config'].get('mode','')[:60]))"; }
run() { # name frames env... -- extra args
  name=$1; mb=$2; shift 2
  env "$@" python bench.py --minibatch $mb --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline $EXTRA 2>/dev/null | grep "^{" | tail -1 | line $name $mb >> $OUT
}
for r in 1 2; do
  for mb in 128 512 1024 2048 4096; do
    EXTRA="" run shipped $mb GRL_RESERVED_CUS=0
    EXTRA="" run reserved1 $mb GRL_RESERVED_CUS=1
    EXTRA="--critic-gate edge0" run reserved1_gate_always $mb GRL_RESERVED_CUS=1
  done
done
cat $OUT
