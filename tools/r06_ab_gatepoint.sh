#!/bin/bash
# On the GPU box: where the critic's lane starts at 4096 frames -- behind the first edge convolution (default) or behind the first fiber convolution.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_gatepoint.txt
: > $OUT
run() { local label=$1 wl=$2; shift 2
  GRL_BENCH_NO_SELFCHECK=1 python bench.py --workload $wl --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline --no-cpu-baseline --repeats 5 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %-12s: %8.2f steps/s  %.4f ms/step  (min %.4f)' % ('$label', '$wl', d['value'], d['ms_per_step'], d['ms_per_step_min_max'][0]))" >> $OUT
}
for round in 1 2 3; do
  run edge0 rigid_hepi
  run fiber0 rigid_hepi --critic-gate fiber0
done
for round in 1 2; do
  run edge0 cloth_hepi
  run fiber0 cloth_hepi --critic-gate fiber0
done
cat $OUT
