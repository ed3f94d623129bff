#!/bin/bash
# On the GPU box (round 6, call e): GPU tests of the new paths (full tail), then the two recorded forms of run_minibatches and the balanced
# node order, alternating rounds on one box.
TAG=${1:-r06e}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_oneshot.py tests/test_gpu_rollout.py tests/test_gpu_step.py tests/test_gpu_determinism.py tests/test_gpu_fullsize.py tests/test_gpu_reference_golden.py -q 2>&1 | tail -60 > gpurun_out/gpu_new_$TAG.txt
tail -25 gpurun_out/gpu_new_$TAG.txt
OUT=gpurun_out/ab_forms_$TAG.txt
: > $OUT
run() {  # label, minibatch, extra args (env via GRL_ENVS)
  local label=$1 mb=$2; shift 2
  env $GRL_ENVS GRL_BENCH_NO_SELFCHECK=1 python bench.py --minibatch $mb --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline --no-cpu-baseline --repeats 5 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %5d frames: %8.2f steps/s  %.4f ms/step  (min %.4f)  host %.4f  %s' % ('$label', $mb, d['value'], d['ms_per_step'], d['ms_per_step_min_max'][0], d['host_enqueue_ms_per_step'], d['mode'][7:60]))" >> $OUT
}
for round in 1 2; do
  for mb in 256 512 1024 4096; do
    GRL_ENVS="GRL_BALANCE_NODE_ORDER=0" run perstep_natural $mb --unroll 1
    GRL_ENVS="GRL_X=0" run perstep_balanced $mb --unroll 1
    GRL_ENVS="GRL_X=0" run default $mb
    GRL_ENVS="GRL_BALANCE_NODE_ORDER=0" run default_natural $mb
  done
  GRL_ENVS="GRL_X=0" run default 32
done
cat $OUT
cd /tmp && export TMPDIR=/tmp
for mb in 512; do
  O=$GRAFT_REPO_ROOT/gpurun_out/tl_${TAG}_$mb
  rocprofv3 --kernel-trace --output-format csv -d $O -o g -- python3 $GRAFT_REPO_ROOT/bench.py --minibatch $mb --steps 24 --warmup 8 --pool 16 --no-cpu-baseline --no-roofline --no-parity-gate --repeats 3 > /dev/null 2>&1
  f=$(find $O -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $f > $GRAFT_REPO_ROOT/gpurun_out/timeline_${TAG}_$mb.txt 2>&1
  rm -rf $O
done
