"""Timeline of ONE replayed policy-update step from a rocprofv3 kernel trace (…_kernel_trace.csv): start (us, relative), duration, gap to the
previous kernel's end, kernel name -- and the busy / idle split of the step.  Steps are delimited by the minibatch gather launch.
   python tools/timeline.py gpurun_out/<dir>/<name>_kernel_trace.csv [step_index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46])
            for r in rows)
# a step starts at the minibatch gather (bench.py: one grl_gather_rows_many launch per step, outside the recorded graph); older traces
# without it: at the launch behind an Adam launch (one Adam per step until round 3, two -- critic's lane, actor's lane -- since)
starts = [i for i, e in enumerate(ev) if "gather_rows_many" in e[2]]
if any("step_head_kernel" in e[2] for e in ev):
    # round 6: with several steps per launch each lane gathers its own inputs (two gathers per step, the critic's mid-step): a step then starts
    # at the actor lane's gather = the gather launch directly in front of the merged head launch
    heads = [i for i, e in enumerate(ev) if "step_head_kernel" in e[2]]
    starts = [max([g for g in starts if g < h], default=h) for h in heads]
if len(starts) < 3:
    starts = [i + 1 for i, e in enumerate(ev) if "adam_dev" in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) * 2 // 3
a, b = starts[k], starts[k + 1]
seg = ev[a:b]
t0 = seg[0][0]
busy, cs, ce = 0, seg[0][0], seg[0][1]
for s, e, _ in seg[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = seg[-1][1] - t0
# the step's period on the device (start of this step's first kernel to the start of the next step's) and the idle time between steps
lo, hi = max(1, k - 4), min(len(starts) - 2, k + 4)
periods = [(ev[starts[i + 1]][0] - ev[starts[i]][0]) / 1e3 for i in range(lo, hi + 1)]
between = [(ev[starts[i + 1]][0] - max(e[1] for e in ev[starts[i]:starts[i + 1]])) / 1e3 for i in range(lo, hi + 1)]
print(f"steps {lo}..{hi}: period {sum(periods) / len(periods):.1f} us (min {min(periods):.1f}, max {max(periods):.1f}), idle between steps "
      f"{sum(between) / len(between):.1f} us")
print(f"step {k}: {len(seg)} kernels, span {span / 1e3:.1f} us, device busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us")
prev = t0
for s, e, n in seg:
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:7.1f}  {n}")
    prev = max(prev, e)
