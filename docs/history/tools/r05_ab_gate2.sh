#!/bin/bash
cd $GRAFT_REPO_ROOT
for mb in 4096 1024 512; do
  for round in 1 2; do
    for flag in "" "--critic-gate fwd_end" "--no-critic-gate"; do
      timeout 240 python bench.py --minibatch $mb --pool 16 --no-cpu-baseline --no-roofline --no-parity-gate --repeats 5 $flag 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('rigid $mb'.ljust(12), ('$flag' or 'edge0').ljust(24), round(l['value'],2), round(l['ms_per_step'],4))"
    done
  done
done
