#!/bin/bash
# On the GPU box: GPU suite of the merged-launch tree, then the replayed step with the round-6 merges on / off (alternating rounds, one box), then timelines.
TAG=${1:-r06b}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/gpu_suite_$TAG.txt
cat gpurun_out/gpu_suite_$TAG.txt
OUT=gpurun_out/ab_merge_$TAG.txt
: > $OUT
run() {  # label, minibatch, env...
  local label=$1 mb=$2; shift 2
  env "$@" GRL_BENCH_NO_SELFCHECK=1 python bench.py --minibatch $mb --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline --no-cpu-baseline --repeats 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-14s %5d frames: %8.2f steps/s  %.4f ms/step  (min %.4f)' % ('$label', $mb, d['value'], d['ms_per_step'], d['ms_per_step_min_max'][0]))" >> $OUT
}
for round in 1 2; do
  for mb in 32 512 4096; do
    run r05form     $mb GRL_FUSE_HEAD=0 GRL_FUSE_TAIL_PRE=0 GRL_SIGNAL_IN_KERNEL=0
    run merged      $mb GRL_X=0
    run merged+first $mb GRL_CRITIC_FIRST=1
    run merged+prio $mb GRL_CRITIC_PRIO=normal
    run m+first+prio $mb GRL_CRITIC_FIRST=1 GRL_CRITIC_PRIO=normal
  done
done
cat $OUT
cd /tmp && export TMPDIR=/tmp
for v in merged first; do
  for mb in 32 512; do
    O=$GRAFT_REPO_ROOT/gpurun_out/tl_${TAG}_${v}_$mb
    if [ $v = first ]; then export GRL_CRITIC_FIRST=1; else unset GRL_CRITIC_FIRST; fi
    rocprofv3 --kernel-trace --output-format csv -d $O -o g -- python3 $GRAFT_REPO_ROOT/bench.py --minibatch $mb --steps 20 --warmup 4 --pool 8 --no-cpu-baseline --no-roofline --no-parity-gate > /dev/null 2>&1
    f=$(find $O -name "*kernel_trace.csv" | head -1)
    python3 $GRAFT_REPO_ROOT/tools/timeline.py $f > $GRAFT_REPO_ROOT/gpurun_out/timeline_${TAG}_${v}_$mb.txt 2>&1
    rm -rf $O
  done
done
