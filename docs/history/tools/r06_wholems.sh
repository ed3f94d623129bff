#!/bin/bash
# On the GPU box: the whole-millisecond mode of the long-step workloads.  For each rope workload, several PROCESSES of
# each form of the step: two-lane graphs (default) | one-stream graph | eager launches | two alternating recordings.  Prints per-repeat ms.
TAG=${1:-r06w}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/wholems_$TAG.txt
: > $OUT
run() {  # label, workload, extra args..., env via GRL_ENVS
  local label=$1 wl=$2; shift 2
  env $GRL_ENVS GRL_BENCH_NO_SELFCHECK=1 python bench.py --workload $wl --steps 20 --warmup 4 --pool 8 --no-parity-gate --no-roofline --no-cpu-baseline --repeats 6 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %-12s %8.3f ms  repeats %s  host %.3f' % ('$wl', '$label', d['ms_per_step'], ' '.join('%.3f' % x for x in d['repeats_ms_per_step']), d.get('host_enqueue_ms_per_step', -1)))" >> $OUT
}
for wl in ${GRL_WL:-rope_hepi_bf16 rope_hepi_var}; do
  for proc in 1 2 3 4; do
    GRL_ENVS="GRL_X=0" run two_lanes $wl
    GRL_ENVS="GRL_X=0" run one_stream $wl --one-stream
    GRL_ENVS="GRL_X=0" run no_gate $wl --no-critic-gate
    GRL_ENVS="GRL_GRAPH_COPIES=2" run two_copies $wl
    if [ $proc -le 2 ]; then GRL_ENVS="GRL_X=0" run eager $wl --no-graph; fi
  done
done
cat $OUT
