"""Per-launch medians over the REPLAYED policy-update steps of a rocprofv3 kernel trace: steps are delimited by the Adam launch; steps with the
most common launch count are the replayed ones; for every launch position in the step the median duration over those steps.
   python tools/timeline_stats.py gpurun_out/<dir>/<name>_kernel_trace.csv [other_trace.csv]     (two traces: side by side, A/B)"""
import csv, sys
from collections import Counter


def load(path):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                 r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]) for r in rows)
    adam = [i for i, e in enumerate(ev) if "adam_dev" in e[2]]
    steps = [ev[a + 1:b + 1] for a, b in zip(adam[:-1], adam[1:])]
    n = Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == n]
    steps = steps[len(steps) // 4:]          # the later ones: warm, replayed
    # launches are keyed by (kernel name, occurrence within the step): concurrent streams may interleave differently from step to step
    keyed = []
    for s in steps:
        seen, d = Counter(), {}
        for st, en, name in s:
            d[(name, seen[name])] = (en - st) / 1e3
            seen[name] += 1
        keyed.append(d)
    keys = list(keyed[0])
    keyed = [d for d in keyed if set(d) == set(keys)]
    med = lambda xs: sorted(xs)[len(xs) // 2]
    dur = {k: med([d[k] for d in keyed]) for k in keys}
    span = med([(s[-1][1] - s[0][0]) / 1e3 for s in steps])
    return keys, dur, span, len(keyed)


a = load(sys.argv[1])
b = load(sys.argv[2]) if len(sys.argv) > 2 else None
print(f"{a[3]} replayed steps of {len(a[0])} launches, median span {a[2]:.1f} us" + (f"   |   {b[3]} steps, span {b[2]:.1f} us" if b else ""))
ta = tb = 0.0
for k in a[0]:
    line = f"{k[0]:42s} #{k[1]} {a[1][k]:8.1f}"
    ta += a[1][k]
    if b and k in b[1]:
        line += f" {b[1][k]:8.1f}  {b[1][k] - a[1][k]:+7.1f}"
        tb += b[1][k]
    print(line)
if b:
    for k in b[0]:
        if k not in a[1]:
            print(f"{k[0]:42s} #{k[1]} {'':8s} {b[1][k]:8.1f}   (second trace only)")
            tb += b[1][k]
print(f"{'sum of launch durations':45s} {ta:8.1f}" + (f" {tb:8.1f}  {tb - ta:+7.1f}" if b else ""))
