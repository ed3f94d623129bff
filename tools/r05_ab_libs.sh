#!/bin/bash
# new Python, every library under _variants/ (the round-4 library included: same ABI), alternating rounds, one box: the replayed two-lane step
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for lib in _variants/lib_*.so; do
    n=$(basename $lib .so); n=${n#lib_}
    GRL_BENCH_NO_SELFCHECK=1 GRL_ALLOW_DIAG_LIB=1 GRL_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-roofline --repeats 5 ${GRL_AB_ARGS:-} 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$n'.ljust(10), round(l['value'],2), round(l['ms_per_step'],4))"
  done
done
