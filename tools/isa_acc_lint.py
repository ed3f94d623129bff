#!/usr/bin/env python3
"""Build-time lint for the asm-pinned MFMA accumulators: the weight-gradient MFMAs of edge_bwd16_kernel and
node_mlp_bwd16_kernel are inline asm with "+a" accumulator tiles, invisible to the compiler's hazard recognizer.  An MFMA result needs
~18 wait states before a vector instruction may read it; the kernels guarantee that by draining (s_nop 15; s_nop 15) and re-defining every
tile behind the drain before the first read.  This lint compiles the files to assembly and FAILS if

  * a v_accvgpr_read / v_accvgpr_mov of a register inside a pinned tile appears within WINDOW instructions AFTER an asm MFMA that writes
    that tile (an in-flight tile read by compiler-generated code), or
  * a kernel that contains asm MFMAs has no drain statement behind the last one, or
  * (round 4) a vector instruction writes a SOURCE register of an asm MFMA less than two wait states before it (the asm statements no
    longer carry their own s_nop: 2-3 % of the two backward kernels' instructions).
(Compiler-generated v_accvgpr moves of pinned tiles far behind the last MFMA -- e.g. behind a loop's closing barrier -- are harmless and
are not flagged; what the drain guarantees is that nothing can be scheduled INTO the window, whatever a future compiler does.)

  python tools/isa_acc_lint.py            (all four objects: fp32 and bf16 builds of both files; exit code 1 on a finding)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WINDOW = 20
FILES = ["edge_conv16.hip", "node_mlp16.hip"]


def compile_s(src, flags, out):
    from geometry_rl_amd import hip
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-S", "--cuda-device-only"] + \
          hip.FILE_FLAGS.get(src, []) + flags + [os.path.join(hip.CSRC, src), "-o", out]
    subprocess.check_call(cmd)


def kernels(path):
    cur, name = None, None
    for l in open(path):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, cur = m.group(1), []
        elif cur is not None:
            cur.append(l.rstrip("\n"))
            if l.strip().startswith("s_endpgm"):
                yield name, cur
                cur = None


def lint_kernel(name, lines):
    findings = []
    insts = []            # (text, in_asm)
    in_asm = False
    for l in lines:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith((";", ".")) or re.match(r"^\.?\w+:", t):
            continue
        insts.append((t, in_asm))
    asm_mfma = [(i, t) for i, (t, a) in enumerate(insts) if a and t.startswith("v_mfma")]
    if not asm_mfma:
        return findings
    tiles = set()
    for i, t in asm_mfma:
        m = re.match(r"v_mfma\S+\s+a\[(\d+):(\d+)\]", t)
        if m:
            tiles.add((int(m.group(1)), int(m.group(2))))
    def tile_of(reg):
        for lo, hi in tiles:
            if lo <= reg <= hi:
                return (lo, hi)
        return None
    def acc_regs(t):
        regs = []
        if t.startswith(("v_accvgpr_read", "v_accvgpr_mov")):
            src = t.split(",")[-1]
            for m in re.finditer(r"a\[(\d+):(\d+)\]|a(\d+)", src):
                regs += list(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else [int(m.group(3))]
        return regs
    # (1) reads shortly after an asm MFMA writing the same tile
    for i, t in asm_mfma:
        m = re.match(r"v_mfma\S+\s+a\[(\d+):(\d+)\]", t)
        if not m:
            continue
        tile = (int(m.group(1)), int(m.group(2)))
        for j in range(i + 1, min(i + 1 + WINDOW, len(insts))):
            tj, aj = insts[j]
            if tj.startswith("s_nop 15"):
                break
            if not aj and any(tile_of(r) == tile for r in acc_regs(tj)):
                findings.append(f"{name}: '{tj}' {j - i} instructions after asm '{t[:60]}'")
    # (3) a source register of an asm MFMA written by a vector instruction less than two wait states before it (the compiler's hazard
    #     recognizer does this for builtin MFMAs -- VALU write -> MFMA SrcA / SrcB read needs 2 wait states on gfx950 -- but not for asm)
    def vregs(tok):
        regs = []
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
            regs += list(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else [int(m.group(3))]
        return regs
    for i, t in asm_mfma:
        ops_ = t.split(None, 1)[1].split(",")
        srcs = set(vregs(ops_[1]) + vregs(ops_[2])) if len(ops_) >= 3 else set()
        ws, j = 0, i - 1
        while j >= 0 and ws < 2:
            tj = insts[j][0]
            mnop = re.match(r"s_nop\s+(\d+)", tj)
            if tj.startswith("v_") and not tj.startswith("v_mfma") and not tj.startswith("v_cmp"):
                dst = tj.split(None, 1)[1].split(",")[0] if " " in tj else ""
                if srcs & set(vregs(dst)):
                    findings.append(f"{name}: '{tj}' writes a source of asm '{t[:70]}' {ws} wait states before it")
            ws += int(mnop.group(1)) + 1 if mnop else 1
            j -= 1
    # (2) the drain
    last = asm_mfma[-1][0]
    drain = next((j for j in range(last, len(insts)) if insts[j][0].startswith("s_nop 15") and insts[j][1]), None)
    if drain is None:
        findings.append(f"{name}: asm MFMAs but no drain (s_nop 15) behind the last one")
    return findings


def main():
    bad = []
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as td:
        jobs = [(src, tag, flags, os.path.join(td, f"{src}.{tag}.s")) for src in FILES for tag, flags in (("fp32", []), ("bf16", ["-DGRL_PREC=1"]))]
        with ThreadPoolExecutor(4) as ex:
            list(ex.map(lambda j: compile_s(j[0], j[2], j[3]), jobs))
        for src, tag, flags, out in jobs:
            if True:
                n = 0
                for name, lines in kernels(out):
                    f = lint_kernel(name, lines)
                    n += 1
                    bad += [f"[{src} {tag}] " + x for x in f]
                print(f"{src} ({tag}): {n} kernels scanned")
    for b in bad:
        print("FINDING:", b)
    print("accumulator lint:", "FAILED" if bad else "clean")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
