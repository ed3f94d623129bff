#!/bin/bash
# On the GPU box: every _variants/lib_*.so at several minibatch sizes and on the other workloads, alternating (one box): ms per step.
cd $GRAFT_REPO_ROOT
for mb in ${GRL_SIZES:-256 512 1024 2048}; do
  for r in 1 2; do for lib in _variants/lib_*.so; do
    GRL_LIB=$PWD/$lib python bench.py --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-8s rigid_hepi minibatch %5d : %8.2f steps/s  %.4f ms/step' % ('$lib'.split('lib_')[1][:-3], $mb, d['value'], d['ms_per_step']))"
  done; done
done
for wl in ${GRL_WLS:-cloth_hepi rope_hepi_var rope_hepi_bf16 rigid2_empn}; do
  for r in 1 2; do for lib in _variants/lib_*.so; do
    GRL_LIB=$PWD/$lib python bench.py --workload $wl --steps 30 --warmup 4 --pool 8 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %-16s 4096 frames: %8.2f steps/s  %.3f ms/step' % ('$lib'.split('lib_')[1][:-3], '$wl', d['value'], d['ms_per_step']))"
  done; done
done
