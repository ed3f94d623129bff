#!/bin/bash
# new Python, every library under _variants/: the four MFMA kernels as the replayed step runs them (in-graph stamps) beside the step itself
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for lib in _variants/lib_*.so; do
    n=$(basename $lib .so); n=${n#lib_}
    GRL_BENCH_NO_SELFCHECK=1 GRL_ALLOW_DIAG_LIB=1 GRL_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-parity-gate --repeats 3 ${GRL_AB_ARGS:-} 2>/dev/null | tail -1 | python -c "
import sys,json; l=json.loads(sys.stdin.read()); k=l['roofline']['replayed_launches']; e=l['roofline']['per_kernel_ms_per_step']
print('$n'.ljust(8), round(l['value'],2), round(l['ms_per_step'],4), 'replayed:', ' '.join('%s %.4f' % (n.replace('_kernel','')[:16], v['ms_per_step']) for n,v in k.items()), '| alone:', ' '.join('%s %.4f' % (n.replace('_kernel','')[:14], e[n]) for n in k))"
  done
done
