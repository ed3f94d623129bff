#!/bin/bash
# On the GPU box: every library under _variants/ (same ABI as the tree's) under the tree's Python, alternating rounds on one box, at the sizes in
# GRL_AB_SIZES (default "32 512 4096").   usage: bash tools/r06_ab_libs.sh <tag>
TAG=${1:-r06libs}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_libs_$TAG.txt
: > $OUT
for round in 1 2 3; do
  for mb in ${GRL_AB_SIZES:-32 512 4096}; do
    for lib in _variants/lib_*.so; do
      n=$(basename $lib .so); n=${n#lib_}
      GRL_BENCH_NO_SELFCHECK=1 GRL_ALLOW_DIAG_LIB=1 GRL_LIB=$PWD/$lib python bench.py --minibatch $mb --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline --no-cpu-baseline --repeats 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %5d frames: %8.2f steps/s  %.4f ms/step  (min %.4f)' % ('$n', $mb, d['value'], d['ms_per_step'], d['ms_per_step_min_max'][0]))" >> $OUT
    done
  done
done
cat $OUT
