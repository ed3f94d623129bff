#!/bin/bash
# On the GPU box: the replayed two-lane step under the round-6 lane switches (host enqueue order of the lanes, priority of the critic's stream),
# alternating rounds on ONE box, at shard sizes and at the headline size; then kernel timelines of the candidates.   usage: bash tools/r06_ab_lanes.sh <tag>
TAG=${1:-r06ab}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_lanes_$TAG.txt
: > $OUT
run() {  # label, minibatch, env...
  local label=$1 mb=$2; shift 2
  env "$@" GRL_BENCH_NO_SELFCHECK=1 python bench.py --minibatch $mb --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline --no-cpu-baseline --repeats 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-14s %5d frames: %8.2f steps/s  %.4f ms/step  (min %.4f)' % ('$label', $mb, d['value'], d['ms_per_step'], d['ms_per_step_min_max'][0]))" >> $OUT
}
for round in 1 2; do
  for mb in 32 512 4096; do
    run base        $mb GRL_X=0
    run first       $mb GRL_CRITIC_FIRST=1
    run prio_normal $mb GRL_CRITIC_PRIO=normal
    run first+prio  $mb GRL_CRITIC_FIRST=1 GRL_CRITIC_PRIO=normal
  done
done
cat $OUT
cd /tmp && export TMPDIR=/tmp
for v in first both; do
  for mb in 32 512; do
    O=$GRAFT_REPO_ROOT/gpurun_out/tl_${TAG}_${v}_$mb
    if [ $v = first ]; then export GRL_CRITIC_FIRST=1; unset GRL_CRITIC_PRIO; else export GRL_CRITIC_FIRST=1 GRL_CRITIC_PRIO=normal; fi
    rocprofv3 --kernel-trace --output-format csv -d $O -o g -- python3 $GRAFT_REPO_ROOT/bench.py --minibatch $mb --steps 20 --warmup 4 --pool 8 --no-cpu-baseline --no-roofline --no-parity-gate > /dev/null 2>&1
    f=$(find $O -name "*kernel_trace.csv" | head -1)
    python3 $GRAFT_REPO_ROOT/tools/timeline.py $f > $GRAFT_REPO_ROOT/gpurun_out/timeline_${TAG}_${v}_$mb.txt 2>&1
    rm -rf $O
  done
done
