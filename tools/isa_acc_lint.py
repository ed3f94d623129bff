#!/usr/bin/env python3
"""Command-line form of the build-time ISA lint (geometry_rl_amd/isa_lint.py; hip.build() runs the same rules on the assembly its own
compile leaves behind and fails the build on a finding).  Compiles edge_conv16.hip and node_mlp16.hip (fp32 and bf16 builds) to assembly
with the library's flags and prints the findings.

  python tools/isa_acc_lint.py            (exit code 1 on a finding)
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compile_s(src, flags, out):
    from geometry_rl_amd import hip
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-S", "--cuda-device-only"] + \
          hip.FILE_FLAGS.get(src, []) + flags + [os.path.join(hip.CSRC, src), "-o", out]
    subprocess.check_call(cmd)


def main():
    from concurrent.futures import ThreadPoolExecutor
    from geometry_rl_amd import isa_lint
    bad = []
    with tempfile.TemporaryDirectory() as td:
        jobs = [(src, tag, flags, os.path.join(td, f"{src}.{tag}.s")) for src in isa_lint.FILES for tag, flags in (("fp32", []), ("bf16", ["-DGRL_PREC=1"]))]
        with ThreadPoolExecutor(4) as ex:
            list(ex.map(lambda j: compile_s(j[0], j[2], j[3]), jobs))
        for src, tag, flags, out in jobs:
            n, n_asm, f = isa_lint.lint_assembly(out)
            bad += [f"[{src} {tag}] " + x for x in f]
            if n_asm == 0:
                bad.append(f"[{src} {tag}] no kernel with asm MFMAs found")
            print(f"{src} ({tag}): {n} kernels scanned, {n_asm} with asm MFMAs")
    for b in bad:
        print("FINDING:", b)
    print("accumulator lint:", "FAILED" if bad else "clean")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
