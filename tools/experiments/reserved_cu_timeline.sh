#!/bin/bash
# On the GPU box: timelines of the multi-step launch with the gate inside (GRL_RESERVED_CUS=1) against the shipped form: which launches pay?
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl_reserved
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mb in 512 4096; do
  for r in 0 1; do
    export GRL_RESERVED_CUS=$r
    rocprofv3 --kernel-trace --output-format csv -d $OUT/p${mb}_$r -o g -- python3 $GRAFT_REPO_ROOT/bench.py --minibatch $mb --steps 24 --warmup 8 --pool 8 --no-cpu-baseline --no-roofline --no-parity-gate > /dev/null 2>&1
    f=$(find $OUT/p${mb}_$r -name "*kernel_trace.csv" | head -1)
    cp $f $OUT/trace_${mb}_$r.csv
    python3 $GRAFT_REPO_ROOT/tools/timeline.py $f > $OUT/timeline_${mb}_$r.txt 2>&1
    rm -rf $OUT/p${mb}_$r
  done
  python3 $GRAFT_REPO_ROOT/tools/timeline_diff.py $OUT/trace_${mb}_0.csv $OUT/trace_${mb}_1.csv > $OUT/diff_$mb.txt 2>&1
  gzip -f $OUT/trace_${mb}_0.csv $OUT/trace_${mb}_1.csv
done
cat $OUT/diff_4096.txt
