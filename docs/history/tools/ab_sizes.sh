#!/bin/bash
# A/B on ONE box at several minibatch sizes: the committed baseline copy under _ab/base (tools/ab_prepare.sh [rev]) vs the working tree,
# alternating, ms per step (and optional extra workloads).   usage: bash tools/ab_sizes.sh "32 512 4096" ["cloth_hepi rigid2_empn"]
cd $GRAFT_REPO_ROOT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-5s %-14s %5s : %8.2f steps/s  %.4f ms/step' % ('$1', '$2', '$3', d['value'], d['ms_per_step']))"; }
for mb in ${1:-32 512 4096}; do
  for r in 1 2; do
    (cd _ab/base && python bench.py --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | line base rigid_hepi $mb)
    python bench.py --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | line new rigid_hepi $mb
  done
done
for wl in $2; do
  for r in 1 2; do
    (cd _ab/base && python bench.py --workload $wl --steps 30 --warmup 4 --pool 8 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | line base $wl 4096)
    python bench.py --workload $wl --steps 30 --warmup 4 --pool 8 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | line new $wl 4096
  done
done
