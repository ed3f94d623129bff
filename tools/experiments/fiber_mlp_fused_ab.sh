#!/bin/bash
# On the GPU box: the fused fiber convolution + ConvNeXt forward (GRL_FUSE_FIBER_MLP) -- its tests, then the step with and without, alternating.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_fiber_mlp.txt
: > $OUT
timeout 900 python -m pytest tests/test_gpu_fiber_mlp_fused.py -x -q 2>&1 | tail -15 | tee -a $OUT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %5s : %8.2f steps/s  %.4f ms/step' % ('$1', '$2', d['value'], d['ms_per_step']))"; }
for r in 1 2 3; do
  for mb in 512 4096; do
    for f in 0 1; do
      GRL_FUSE_FIBER_MLP=$f python bench.py --minibatch $mb --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | line fuse$f $mb >> $OUT
    done
  done
done
for f in 0 1; do
  GRL_FUSE_FIBER_MLP=$f python bench.py --workload rope_hepi_bf16 --steps 20 --warmup 5 --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | line bf16_fuse$f 4096 >> $OUT
done
cat $OUT
