#!/bin/bash
# A/B on one box: the critic's lane gated behind the actor's first edge convolution (default) vs starting with the step (--no-critic-gate)
cd $GRAFT_REPO_ROOT
for wl_mb in "rigid_hepi 4096" "rigid_hepi 2048" "rigid_hepi 1024" "rigid_hepi 512" "rigid_hepi 32" "cloth_hepi 4096" "rigid2_empn 4096" "rope_hepi_bf16 4096"; do
  set -- $wl_mb
  for round in 1 2; do
    for flag in "" "--no-critic-gate"; do
      timeout 240 python bench.py --workload $1 --minibatch $2 --pool 16 --no-cpu-baseline --no-roofline --no-parity-gate --repeats 5 $flag 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$1 $2'.ljust(22), ('gate' if '$flag'=='' else 'nogate').ljust(7), round(l['value'],2), round(l['ms_per_step'],4))"
    done
  done
done
