"""The round's measuring tools run on CPU inputs: tools/isa_audit.py on the assembly hip.build() keeps for the ISA lint, tools/timeline_diff.py
on two synthetic kernel traces."""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_isa_audit_reads_the_builds_assembly():
    from geometry_rl_amd import hip
    hip.build(verbose=False)
    files = sorted(glob.glob(os.path.join(ROOT, "geometry_rl_amd", "csrc", "build", "node_mlp16.hip*.d", "*gfx950.s")))
    assert files, "hip.build() keeps the assembly of the files with asm MFMAs (-save-temps) for the ISA lint"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_audit.py")] + files, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    loops = [l for l in out.stdout.splitlines() if " loop " in l]
    assert any("node_mlp_bwd16" in l for l in loops), out.stdout[:2000]          # the kernel's chunk loop is found ...
    assert any("v_mfma" in l or "v_pk" in l or "v_" in l for l in loops)         # ... with an instruction mix behind it


def _trace(path, durs):
    """a kernel trace of 12 identical steps: gather, a, b with the given durations (us), back to back"""
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writeheader()
        t = 1_000_000
        for _ in range(12):
            for name, d in (("gather_rows_many_kernel(x)", 5.0), ("(anonymous namespace)::a_kernel(int)", durs[0]), ("void b_kernel<0>(float*)", durs[1])):
                w.writerow({"Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + int(d * 1000)})
                t += int(d * 1000)


def test_timeline_diff_finds_the_launch_that_pays(tmp_path):
    a, b = str(tmp_path / "a.csv"), str(tmp_path / "b.csv")
    _trace(a, (100.0, 700.0))
    _trace(b, (87.0, 715.0))     # the first kernel faster, the one behind it slower: the pattern of profiles/r05_tl_diff_fiber_pk.txt
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "timeline_diff.py"), a, b], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = {l.split()[0]: l for l in out.stdout.splitlines() if "dur" in l}
    assert "-13.0" in lines["a_kernel"] and "+15.0" in lines["b_kernel<0>"], out.stdout
    assert "sum of duration differences +2.0 us" in out.stdout
