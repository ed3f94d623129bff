#!/bin/bash
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for lib in r04 now; do
    GRL_BENCH_NO_SELFCHECK=1 GRL_ALLOW_DIAG_LIB=1 GRL_LIB=$PWD/_variants/lib_$lib.so python bench.py --no-cpu-baseline --no-parity-gate --repeats 3 2>/dev/null | tail -1 | python -c "
import sys,json; l=json.loads(sys.stdin.read()); k=l['roofline']['per_kernel_ms_per_step']
print('newpy+$lib', round(l['value'],2), round(l['ms_per_step'],4), ' '.join('%s %.4f' % (n.replace('grl_','').replace('_kernel','')[:18], v) for n,v in list(k.items())[:14]))"
  done
done
