#!/bin/bash
# two libraries under _variants/ ($1 $2): rocprofv3 kernel trace of the replayed 4096-frame step with each, then tools/timeline_diff.py
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tld; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp GRL_ALLOW_DIAG_LIB=1 GRL_BENCH_NO_SELFCHECK=1
for n in $1 $2 $1 $2; do
  export GRL_LIB=$R/_variants/lib_$n.so
  rm -rf $OUT/p_$n
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/p_$n -o t -- python3 $R/bench.py --minibatch ${GRL_TL_MB:-4096} --steps 30 --warmup 6 --pool 8 --no-cpu-baseline --no-roofline --no-parity-gate > /dev/null 2>&1
  cp $(find $OUT/p_$n -name "*kernel_trace.csv" | head -1) $OUT/trace_$n.csv
done
python3 $R/tools/timeline_diff.py $OUT/trace_$1.csv $OUT/trace_$2.csv
