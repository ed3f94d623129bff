#!/bin/bash
# HSA_ENABLE_INTERRUPT=0 (the ROCr runtime busy-polls its signals instead of sleeping on interrupts) against the default, alternating, per workload:
# the long-step workloads read k.000 ms per step on some boxes / in some repeats (rope_hepi_var 12.25 or 13.0) -- a 1-ms wake-up somewhere in the runtime?
cd $GRAFT_REPO_ROOT
for wl in ${GRL_AB_WORKLOADS:-rigid_hepi rope_hepi rope_hepi_bf16}; do
  for round in $(seq 1 ${GRL_AB_ROUNDS:-3}); do
    for mode in default poll; do
      if [ $mode = poll ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
      timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-parity-gate --repeats 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$wl', '$mode'.ljust(8), round(d['value'],2), round(d['ms_per_step'],4), [round(x,3) for x in d['repeats_ms_per_step']])"
    done
  done
done
