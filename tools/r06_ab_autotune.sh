#!/bin/bash
# On the GPU box: the measured choice of the recorded form (GRL_AUTOTUNE_FORM) -- its test, then what it picks per workload and size and what
# the step then costs, against the table alone.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_autotune.txt
: > $OUT
timeout 900 python -m pytest tests/test_gpu_rollout.py -x -q 2>&1 | tail -4 | tee -a $OUT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); lf=d.get('lane_form') or {}; print('%-14s %-9s %5s : %8.2f steps/s  %.4f ms/step  form %-9s measured unrolled %.4f per_step %.4f' % ('$1', '$2', '$3', d['value'], d['ms_per_step'], lf.get('form','(table)'), lf.get('unrolled_ms_per_step',0), lf.get('per_step_ms_per_step',0)))"; }
run() { w=$1; name=$2; mb=$3; shift 3
  python bench.py --workload $w --minibatch $mb --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline --no-cpu-baseline "$@" 2>/dev/null | grep "^{" | tail -1 | line $w $name $mb >> $OUT
}
for r in 1 2; do
  for w in rigid_hepi cloth_hepi rigid2_empn rope_hepi_bf16; do
    for mb in 512 4096; do
      GRL_AUTOTUNE_FORM=1 run $w measured $mb
      GRL_AUTOTUNE_FORM=0 run $w table $mb
    done
  done
  GRL_AUTOTUNE_FORM=1 run rigid_hepi measured 128
  GRL_AUTOTUNE_FORM=0 run rigid_hepi table 128
  GRL_AUTOTUNE_FORM=1 run rigid_hepi measured 2048
  GRL_AUTOTUNE_FORM=0 run rigid_hepi table 2048
done
cat $OUT
