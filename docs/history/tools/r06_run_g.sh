#!/bin/bash
# On the GPU box (round 6, call g): the two tests that failed in call f, then the UNGATED unrolled form (critic's lane free-running inside a
# multi-step launch) against the default at the gated sizes, alternating rounds on one box; then one rope line (whole-millisecond detector).
TAG=${1:-r06g}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_rollout.py tests/test_gpu_oneshot.py -q 2>&1 | tail -30 > gpurun_out/gpu_new_$TAG.txt
tail -8 gpurun_out/gpu_new_$TAG.txt
OUT=gpurun_out/ab_ungated_$TAG.txt
: > $OUT
run() {  # label, minibatch, extra args (env via GRL_ENVS)
  local label=$1 mb=$2; shift 2
  env $GRL_ENVS GRL_BENCH_NO_SELFCHECK=1 python bench.py --minibatch $mb --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline --no-cpu-baseline --repeats 5 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-18s %5d frames: %8.2f steps/s  %.4f ms/step  (min %.4f)  host %.4f  %s' % ('$label', $mb, d['value'], d['ms_per_step'], d['ms_per_step_min_max'][0], d['host_enqueue_ms_per_step'], d['mode'][7:60]))" >> $OUT
}
for round in 1 2; do
  for mb in 128 256 512 2048 4096; do
    GRL_ENVS="GRL_X=0" run default $mb
    GRL_ENVS="GRL_X=0" run ungated_unroll8 $mb --no-critic-gate
    GRL_ENVS="GRL_X=0" run ungated_unroll2 $mb --no-critic-gate --unroll 2
  done
done
cat $OUT
GRL_BENCH_NO_SELFCHECK=1 python bench.py --workload rope_hepi_var --steps 20 --warmup 4 --pool 8 --no-parity-gate --no-roofline --no-cpu-baseline --repeats 6 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('rope_hepi_var %.3f ms  repeats %s' % (d['ms_per_step'], ' '.join('%.3f' % x for x in d['repeats_ms_per_step'])))" | tee gpurun_out/rope_probe_$TAG.txt
