#!/bin/bash
# On the GPU box: what does the critic's lane cost the actor at 4096 frames?  Inside an UNGATED eight-step launch the critic runs ahead and
# the later steps of the launch are the actor alone: their span against the gated per-step program's step.
OUT=$GRAFT_REPO_ROOT/gpurun_out/actor_alone
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GRL_AUTOTUNE_FORM=0
for form in unrolled per_step; do
  if [ $form = unrolled ]; then export GRL_EPOCH_GATED_FROM=100000000; else unset GRL_EPOCH_GATED_FROM; fi
  rocprofv3 --kernel-trace --output-format csv -d $OUT/p_$form -o g -- python3 $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 8 --pool 8 --no-cpu-baseline --no-roofline --no-parity-gate > /dev/null 2>&1
  f=$(find $OUT/p_$form -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/timeline.py $f > $OUT/timeline_$form.txt 2>&1
  python3 - $f > $OUT/steps_$form.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# actor steps: from step_head_kernel to the next step_head_kernel (actor lane kernels only: names not starting with ds_ / value_loss / build_features / wait_flag)
heads = [i for i, e in enumerate(ev) if "step_head_kernel" in e[2]]
tails = [e for e in ev if "reduce_partials_multi_kernel<true>" in e[2] or "reduce_partials_multi_kernel<(bool)1>" in e[2]]
spans = []
for a, b in zip(heads[:-1], heads[1:]):
    t0 = ev[a][0]
    # the actor's tail fold of this step: the LAST reduce_partials_multi before the next head
    ends = [e[1] for e in ev[a:b] if "reduce_partials_multi" in e[2]]
    if ends:
        spans.append((max(ends) - t0) / 1e3)
print("actor step spans (us), last 40:", [round(x, 1) for x in spans[-40:]])
print("min", round(min(spans[-40:]), 1), "median", round(sorted(spans[-40:])[20], 1))
PY
  rm -rf $OUT/p_$form
done
cat $OUT/steps_unrolled.txt $OUT/steps_per_step.txt
