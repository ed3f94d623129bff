#!/bin/bash
# the critic gate on the long-step workloads (rope): with / without, alternating -- do the exact-millisecond step times (17.000, 13.000 ms) come from the
# hipStreamWaitValue32 in front of the critic's lane?
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for wl in ${GRL_AB_WORKLOADS:-rope_hepi_var rope_hepi_bf16 rope_hepi}; do
    for gate in "" "--no-critic-gate"; do
      timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-parity-gate --repeats 5 $gate 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$wl', '${gate:-gate}'.ljust(16), round(d['value'],2), round(d['ms_per_step'],4), [round(x,3) for x in d['repeats_ms_per_step']])"
    done
  done
done
