#!/bin/bash
# On the GPU box: kernel-by-kernel timelines of one replayed step at 32 / 512 / 4096 frames (rocprofv3 kernel trace + tools/timeline.py)
# and the minibatch sweep.   usage: bash tools/prof_timelines.sh <tag>
TAG=${1:-run}
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mb in ${GRL_TL_SIZES:-32 512 4096}; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/p$mb -o g$mb -- python3 $GRAFT_REPO_ROOT/bench.py --minibatch $mb --steps 20 --warmup 4 --pool 8 --no-cpu-baseline --no-roofline --no-parity-gate > /dev/null 2>&1
  f=$(find $OUT/p$mb -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $f > $OUT/timeline_$mb.txt 2>&1
  rm -rf $OUT/p$mb
done
cd $GRAFT_REPO_ROOT
for mb in 32 256 512 1024 2048 4096; do
  python bench.py --minibatch $mb --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('rigid_hepi minibatch %5d : %8.2f steps/s  %.3f ms/step  mode %s' % ($mb, d['value'], d['ms_per_step'], d['mode']))"
done > $OUT/sizes.txt 2>&1
cat $OUT/sizes.txt
