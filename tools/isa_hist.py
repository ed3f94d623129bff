#!/usr/bin/env python3
"""Instruction mix of a kernel's hot loop from hipcc's assembly listing (run HERE; no GPU).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s file.hip
  python tools/isa_hist.py k.s edge_bwd16_kernel [--loop N | --whole] [--list]

The hot loop is taken to be the LONGEST backward-branch body of the kernel (label .. s_cbranch to that label); --loop picks the
N-th longest, --whole takes the whole kernel.  Categories carry the issue prices measured with tools/ubench/valu_rates.hip
(profiles/r02_valu_rates.txt; cycles per wave instruction when the vector pipe is saturated): plain VALU 2.5 (5.0 for a lone wave),
packed f32 4.45, transcendental 8.5, cvt_pk / perm 4.4; MFMA pipe time 16 cycles for 16x16x32 bf16 and 32 for 32x32x16.
"""
import re
import sys
from collections import Counter

TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_")
HALF = ("v_cvt_pk_bf16_f32", "v_perm_b32", "v_cvt_pk")


def category(op):
    if op.startswith("v_mfma"):
        return "mfma32" if "32x32" in op else "mfma16"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith(HALF):
        return "valu_half"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_rd"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "lds_wr"
    if op.startswith("ds_"):
        return "lds_other"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "scratch_rd" if op.startswith("scratch") else "vmem_rd"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "scratch_wr" if op.startswith("scratch") else "vmem_wr"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


PRICE_SAT = {"valu": 2.5, "valu_pk": 4.45, "trans": 8.5, "valu_half": 4.4, "accvgpr": 2.5, "lane": 2.5}
PRICE_LONE = {"valu": 5.0, "valu_pk": 5.2, "trans": 8.5, "valu_half": 5.0, "accvgpr": 5.0, "lane": 5.0}
MFMA_CYC = {"mfma16": 16, "mfma32": 32}


def kernel_lines(path, name):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*" + re.escape(name) + r"\w*:", l):
            start = i
            break
    if start is None:
        raise SystemExit(f"kernel {name} not found")
    out = []
    for l in lines[start + 1:]:
        out.append(l)
        if l.strip().startswith("s_endpgm"):
            break
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    whole = "--whole" in sys.argv
    nth = int(sys.argv[sys.argv.index("--loop") + 1]) if "--loop" in sys.argv else 0
    ls = kernel_lines(path, name)
    labels = {}
    for i, l in enumerate(ls):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(ls):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    loops.sort(reverse=True)
    if whole or not loops:
        lo, hi = 0, len(ls)
    else:
        _, lo, hi = loops[nth]
    body = [l for l in ls[lo:hi + 1] if re.match(r"^\s+[a-z]", l) and not l.strip().startswith((".", ";"))]
    ops = [l.split()[0] for l in body]
    cats = Counter(category(o.replace("_e32", "").replace("_e64", "")) for o in ops)
    print(f"{name}: {'whole kernel' if whole or not loops else f'loop lines {lo}-{hi}'}  ({len(ops)} instructions; "
          f"{len(loops)} loops, sizes {[s for s, _, _ in loops[:6]]})")
    for k, v in cats.most_common():
        print(f"  {k:10s} {v:6d}")
    sat = sum(PRICE_SAT.get(k, 0) * v for k, v in cats.items())
    lone = sum(PRICE_LONE.get(k, 0) * v for k, v in cats.items())
    mf = sum(MFMA_CYC.get(k, 0) * v for k, v in cats.items())
    print(f"  vector-pipe cycles: {sat:.0f} saturated / {lone:.0f} lone wave;  matrix-pipe cycles: {mf}")
    if "--list" in sys.argv:
        c = Counter(o.replace("_e32", "").replace("_e64", "") for o in ops)
        for k, v in c.most_common(60):
            print(f"    {k:32s} {v}")


if __name__ == "__main__":
    main()
