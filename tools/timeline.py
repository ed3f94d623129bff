"""Timeline of ONE replayed policy-update step from a rocprofv3 kernel trace (…_kernel_trace.csv): start (us, relative), duration, gap to the
previous kernel's end, kernel name -- and the busy / idle split of the step.  Steps are delimited by the Adam launch.
   python tools/timeline.py gpurun_out/<dir>/<name>_kernel_trace.csv [step_index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46])
            for r in rows)
adam = [i for i, e in enumerate(ev) if "adam_dev" in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) * 2 // 3
a, b = adam[k], adam[k + 1]
seg = ev[a + 1:b + 1]
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
print(f"step {k}: {len(seg)} kernels, span {span / 1e3:.1f} us, device busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us")
prev = t0
for s, e, n in seg:
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:7.1f}  {n}")
    prev = max(prev, e)
