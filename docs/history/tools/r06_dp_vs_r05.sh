#!/bin/bash
# On the GPU box: the data-parallel program (one-rank RCCL group) of THIS tree against round 5's tree (_ab/r05, its own library) at 4096 and 1024
# (_ab/r05 = `git archive 89b724b | tar -x -C _ab/r05` + its own build; the copy is scratch and was deleted at the end of round 6)
# frames, alternating, then a kernel timeline of one DP step of each at 4096 frames.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/dp_vs_r05.txt
: > $OUT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %-10s %5s : %8.2f steps/s  %.4f ms/step' % ('$1', '$2', '$3', d['value'], d['ms_per_step']))"; }
for r in 1 2; do
  for mb in 4096 1024; do
    (cd _ab/r05 && python bench.py --dp-plan --minibatch $mb --steps 60 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | line r05 dp $mb) >> $OUT
    python bench.py --dp-plan --minibatch $mb --steps 60 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | line r06 dp $mb >> $OUT
    python bench.py --unroll 1 --minibatch $mb --steps 60 --warmup 8 --pool 16 --no-parity-gate --no-roofline --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | line r06 one_rank $mb >> $OUT
  done
done
cat $OUT
export TMPDIR=/tmp
for t in r05 r06; do
  if [ $t = r05 ]; then D=$GRAFT_REPO_ROOT/_ab/r05; else D=$GRAFT_REPO_ROOT; fi
  O=$GRAFT_REPO_ROOT/gpurun_out/tl_dp_$t
  (cd $D && rocprofv3 --kernel-trace --output-format csv -d $O -o g -- python3 $D/bench.py --dp-plan --minibatch 4096 --steps 12 --warmup 4 --pool 8 --no-cpu-baseline --no-roofline --no-parity-gate --repeats 2 > /dev/null 2>&1)
  f=$(find $O -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $f > $GRAFT_REPO_ROOT/gpurun_out/timeline_dp_${t}_4096.txt 2>&1
  rm -rf $O
done
