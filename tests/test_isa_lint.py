"""Build-time ISA lint of the asm-pinned MFMA accumulators (tools/isa_acc_lint.py): compiles edge_conv16.hip and
node_mlp16.hip (fp32 and bf16 builds) to assembly with the library's own flags and fails on a compiler-generated accumulator read inside
the hazard window of an asm MFMA, or a missing drain.  CPU only (hipcc cross-compiles)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_asm_accumulator_lint_is_clean():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_acc_lint.py")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "accumulator lint: clean" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
