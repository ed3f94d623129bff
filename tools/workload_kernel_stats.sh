#!/bin/bash
# rocprofv3 kernel statistics of three eager steps of each workload named on the command line (default: cloth_hepi rigid2_empn): the step's
# kernels by total time -- where does a workload other than the headline spend its step?
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp GRL_STEPS=3
for wl in ${@:-cloth_hepi rigid2_empn}; do
  export GRL_WORKLOAD=$wl
  rm -rf $R/gpurun_out/wlstats_$wl
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wlstats_$wl -o st -- python3 $R/tools/profile_step.py > $R/gpurun_out/wlstats_$wl.log 2>&1
  echo "== $wl"
  python3 - <<PY
import csv
for r in list(csv.DictReader(open('$R/gpurun_out/wlstats_$wl/st_kernel_stats.csv')))[:22]:
    print(f"{r['Name'].replace('(anonymous namespace)::','')[:52]:52s} {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f} {r['Percentage']:>6s}%")
PY
done
