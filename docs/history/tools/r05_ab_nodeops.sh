#!/bin/bash
# One gpurun call: every library under _variants/ on rope_hepi_bf16 and rigid_hepi (replayed step; the order of the libraries reverses every
# round -- a box that warms up during the call penalises whoever runs last), means at the end; then rocprofv3 kernel statistics of three
# rope_hepi_bf16 steps per library (the kernels' own durations).   GRL_AB_ROUNDS (4), GRL_AB_WORKLOADS, GRL_AB_TESTS=1 runs the node-op tests first
cd $GRAFT_REPO_ROOT
if [ -n "$GRL_AB_TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_rope.py tests/test_gpu_step.py -x -q 2>&1 | tail -3; fi
LIBS=$(ls _variants/lib_*.so); REV=$(ls -r _variants/lib_*.so)
rm -f gpurun_out/ab_nodeops.txt
for wl in ${GRL_AB_WORKLOADS:-rope_hepi_bf16 rigid_hepi}; do
  echo "== $wl"
  for round in $(seq 1 ${GRL_AB_ROUNDS:-4}); do
    if (( round % 2 )); then order=$LIBS; else order=$REV; fi
    for lib in $order; do
      n=$(basename $lib .so); n=${n#lib_}
      GRL_BENCH_NO_SELFCHECK=1 GRL_ALLOW_DIAG_LIB=1 GRL_LIB=$PWD/$lib timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-parity-gate --repeats 3 2>/dev/null | tail -1 | python -c "
import sys,json; l=json.loads(sys.stdin.read()); print('$wl', '$n'.ljust(8), round(l['value'],2), round(l['ms_per_step'],4), l['loss']['loss_objective'], l['loss']['loss_critic'])" | tee -a gpurun_out/ab_nodeops.txt
    done
  done
done
python - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open('gpurun_out/ab_nodeops.txt'):
    w,n,v,ms=l.split()[:4]; d[(w,n)].append(float(ms))
for k,v in sorted(d.items()): print('mean', k[0], k[1].ljust(8), round(sum(v)/len(v),4), 'ms  ->', round(1e3/(sum(v)/len(v)),2), 'steps/s', v)
PY
R=$GRAFT_REPO_ROOT
export GRL_WORKLOAD=${GRL_AB_PROF_WL:-rope_hepi_bf16} GRL_STEPS=3 GRL_ALLOW_DIAG_LIB=1
cd /tmp && export TMPDIR=/tmp
for n in ${GRL_AB_PROF:-base new2}; do
  export GRL_LIB=$R/_variants/lib_$n.so
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/nodeops_$n -o st -- python3 $R/tools/profile_step.py > $R/gpurun_out/nodeops_$n.log 2>&1
  echo "== $n"; python3 - <<PY
import csv
for r in csv.DictReader(open('$R/gpurun_out/nodeops_$n/st_kernel_stats.csv')):
    if float(r["Percentage"]) > 0.8:
        print(f"{r['Name'].replace('(anonymous namespace)::','')[:44]:44s} {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
done
