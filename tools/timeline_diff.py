"""Two rocprofv3 kernel traces of the same replayed step (e.g. two libraries on one box): per launch position of the step, the kernel's mean
duration and mean start offset over the traced steps, side by side -- which launch pays when a change elsewhere makes the step slower?
   python tools/timeline_diff.py a_kernel_trace.csv b_kernel_trace.csv"""
import csv, sys
def load(path):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]) for r in rows)
    starts = [i for i, e in enumerate(ev) if "gather_rows_many" in e[2]]
    steps = [ev[starts[i]:starts[i + 1]] for i in range(len(starts) * 1 // 3, len(starts) - 1)]
    n = min(len(s) for s in steps)
    steps = [s for s in steps if len(s) == n] if all(len(s) == n for s in steps) else [s[:n] for s in steps]
    out = []
    for k in range(n):
        names = {s[k][2] for s in steps}
        dur = sum(s[k][1] - s[k][0] for s in steps) / len(steps) / 1e3
        off = sum(s[k][0] - s[0][0] for s in steps) / len(steps) / 1e3
        out.append(("/".join(sorted(names)), dur, off))
    period = (steps[-1][0][0] - steps[0][0][0]) / (len(steps) - 1) / 1e3
    return out, period, len(steps)
a, pa, na = load(sys.argv[1]); b, pb, nb = load(sys.argv[2])
print(f"period {pa:.1f} us ({na} steps)  vs  {pb:.1f} us ({nb} steps)")
# align by kernel name (launch order on two lanes may interleave differently): k-th occurrence of each name
from collections import defaultdict
ib = defaultdict(list)
for n_, d, o in b: ib[n_].append((d, o))
seen = defaultdict(int)
tot = 0.0
for n_, d, o in a:
    j = seen[n_]; seen[n_] += 1
    if j < len(ib[n_]):
        d2, o2 = ib[n_][j]
        tot += d2 - d
        flag = "  <--" if abs(d2 - d) > max(3.0, 0.04 * d) else ""
        print(f"{n_[:40]:40s} start {o:8.1f} {o2:8.1f}   dur {d:8.1f} {d2:8.1f}   {d2 - d:+7.1f}{flag}")
print(f"sum of duration differences {tot:+.1f} us")
