#!/bin/bash
# On the GPU box: step time of the bench workloads at several minibatch sizes (the per-GPU share when a 4096-frame minibatch is sharded)
# and the other workloads at 4096 frames.  Prints one line per run.
cd $GRAFT_REPO_ROOT
for mb in 32 128 256 512 1024 2048 4096; do
  python bench.py --minibatch $mb --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('rigid_hepi minibatch %5d : %8.2f steps/s  %.3f ms/step  mode %s' % ($mb, d['value'], d['ms_per_step'], d['mode']))"
done
for wl in cloth_hepi rope_hepi rope_hepi_var rope_hepi_bf16 rigid2_empn; do
  python bench.py --workload $wl --steps 20 --warmup 4 --pool 8 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-16s 4096 frames: %8.2f steps/s  %.3f ms/step  dtype %s' % ('$wl', d['value'], d['ms_per_step'], d['dtype'][:40]))"
done
