#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of the bench for every _variants/lib_*.so, then per-launch medians of the replayed steps side by side
# (first variant against each other one).   gpurun -- 'bash tools/ab_trace.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ARGS=${GRL_VARIANT_ARGS:---steps 100 --warmup 5 --pool 8 --no-parity-gate}
first=""
for lib in _variants/lib_*.so; do
  name=$(basename $lib .so); name=${name#lib_}
  export GRL_LIB=$PWD/$lib GRL_BENCH_NO_PROFILE=1
  rocprofv3 --kernel-trace --output-format csv -d /tmp/abtrace_$name -o t -- python3 bench.py $ARGS > /tmp/abtrace_$name.log 2>&1
  grep -h "^{" /tmp/abtrace_$name.log | cut -c1-120
  f=$(find /tmp/abtrace_$name -name '*kernel_trace.csv' | head -1)
  if [ -z "$first" ]; then first=$f; else echo "== first vs $name"; python3 tools/timeline_stats.py $first $f; fi
done
