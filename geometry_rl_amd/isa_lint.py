"""Build-time lint of the generated gfx950 ISA for the asm-pinned MFMA accumulators (run by ``hip.build()`` on every object it compiles
from a file that contains asm MFMAs; ``tools/isa_acc_lint.py`` is the command-line form).

The weight-gradient MFMAs of ``edge_bwd16_kernel`` and ``node_mlp_bwd16_kernel`` are inline asm with "+a" accumulator tiles
(csrc/grl_tile16.h), invisible to the compiler's hazard recognizer, and since round 4 they carry no ``s_nop`` of their own.  What the
compiler would have guaranteed for a builtin MFMA is checked here on the assembly the real compile leaves behind (``-save-temps``):

  (1) no compiler-generated ``v_accvgpr_read`` / ``v_accvgpr_mov`` FROM a pinned tile within WINDOW instructions after an asm MFMA that
      writes it (an MFMA result needs ~18 wait states before a vector read; the kernels drain with ``s_nop 15; s_nop 15`` first);
  (2) every kernel with asm MFMAs has that drain behind the last one;
  (3) no vector instruction writes an operand of an asm MFMA less than two wait states before the MFMA issues:
        * SrcA / SrcB VGPRs (VALU write -> MFMA read: 2 wait states on gfx950),
        * the accumulator tile itself (``v_accvgpr_write`` / ``v_accvgpr_mov`` INTO the "+a" tile: SrcC, the same hazard),
        * both registers of two-destination instructions (``v_swap_b32``),
      and the scan does not stop at the top of a basic block: it follows the fall-through predecessor and every branch that targets the
      block's label (loop back-edges: a source written at the tail of the previous iteration).
"""
import re

WINDOW = 20
FILES = ("edge_conv16.hip", "node_mlp16.hip")   # the sources with asm MFMAs


def kernels(path):
    """-> (mangled name, [lines]) of every kernel body in an assembly file"""
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


def _regs(tok, bank):
    regs = []
    for m in re.finditer(r"\b%s\[(\d+):(\d+)\]|\b%s(\d+)\b" % (bank, bank), tok):
        regs += list(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else [int(m.group(3))]
    return regs


def _written(t):
    """(vgprs, agprs) a vector instruction writes"""
    if not t.startswith("v_") or t.startswith(("v_cmp", "v_mfma", "v_nop")) or " " not in t:
        return [], []
    ops_ = [o.strip() for o in t.split(None, 1)[1].split(",")]
    dst = ops_[:2] if t.startswith("v_swap") else ops_[:1]
    v, a = [], []
    for d in dst:
        v += _regs(d, "v")
        a += _regs(d, "a")
    return v, a


def lint_kernel(name, lines):
    """-> [findings] (see lint_kernel_ex)"""
    return lint_kernel_ex(name, lines)[0]


def lint_kernel_ex(name, lines):
    """-> ([findings], number of MFMAs found INSIDE ;;#ASMSTART / ;;#ASMEND regions).  Only those count as "asm MFMAs": empty pin
    statements (asm volatile("" : "+v"(x))) emit ASMSTART blocks too, and a kernel whose asm MFMAs vanished (or whose mnemonic changed so
    that nothing below matches) must not pass as "linted, no findings" (ADVICE r5)."""
    findings = []
    insts = []            # (text, in_asm) -- labels are kept as ("<label>:", False)
    in_asm = False
    for l in lines:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.?\w+):", t)
        if m:
            insts.append((m.group(1) + ":", False))
            continue
        if not t or t.startswith((";", ".")):
            continue
        insts.append((t.split(";")[0].strip(), in_asm))
    is_label = lambda i: insts[i][0].endswith(":")
    asm_mfma = [(i, t) for i, (t, a) in enumerate(insts) if a and t.startswith("v_mfma")]
    if not asm_mfma:
        return findings, 0
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

    def acc_read(t):
        if t.startswith(("v_accvgpr_read", "v_accvgpr_mov")):
            return _regs(t.split(",")[-1], "a")
        return []

    branches = {}   # label -> [instruction index of every branch to it]
    label_at = {}   # label -> its instruction index
    for i, (t, _) in enumerate(insts):
        m = re.match(r"s_c?branch\S*\s+(\.?\w+)", t)
        if m:
            branches.setdefault(m.group(1) + ":", []).append(i)
        if is_label(i):
            label_at[t] = i

    # (1) reads shortly after an asm MFMA writing the same tile -- along the fall-through path AND through every branch inside the window
    def scan_down(j, n, tile, mfma_text, seen):
        while j < len(insts) and n < WINDOW:
            if (j, n) in seen:
                return
            seen.add((j, n))
            tj, aj = insts[j]
            if is_label(j):
                j += 1
                continue
            n += 1
            if tj.startswith("s_nop 15") or tj.startswith(("s_endpgm", "s_setpc")):
                return
            if not aj and any(tile_of(r) == tile for r in acc_read(tj)):
                findings.append(f"{name}: '{tj}' {n} instructions after asm '{mfma_text[:60]}'")
            mb = re.match(r"s_(c?)branch\S*\s+(\.?\w+)", tj)
            if mb:
                tgt = label_at.get(mb.group(2) + ":")
                if tgt is not None:
                    scan_down(tgt, n, tile, mfma_text, seen)
                if not mb.group(1):
                    return              # unconditional: no fall-through
            j += 1

    for i, t in asm_mfma:
        m = re.match(r"v_mfma\S+\s+a\[(\d+):(\d+)\]", t)
        if not m:
            continue
        scan_down(i + 1, 0, (int(m.group(1)), int(m.group(2))), t, set())

    # (3) operands written less than two wait states before an asm MFMA, across basic-block boundaries
    def scan_up(j, ws, src_v, src_a, mfma_text, seen):
        """walk upward from instruction j (inclusive) with ``ws`` wait states already between it and the MFMA"""
        while j >= 0 and ws < 2:
            if (j, ws) in seen:
                return
            seen.add((j, ws))
            tj = insts[j][0]
            if is_label(j):
                for b in branches.get(tj, []):      # every branch into this block: the branch itself is one wait state
                    scan_up(b - 1, ws + 1, src_v, src_a, mfma_text, seen)
                prev = insts[j - 1][0] if j > 0 else ""
                if prev.startswith(("s_branch", "s_endpgm", "s_setpc")):
                    return                          # no fall-through into this block
                j -= 1
                continue
            wv, wa = _written(tj)
            if (src_v & set(wv)) or (src_a & set(wa)):
                what = "a source" if (src_v & set(wv)) else "the accumulator tile"
                findings.append(f"{name}: '{tj}' writes {what} of asm '{mfma_text[:70]}' {ws} wait states before it")
            mnop = re.match(r"s_nop\s+(\d+)", tj)
            ws += int(mnop.group(1)) + 1 if mnop else 1
            j -= 1

    for i, t in asm_mfma:
        ops_ = t.split(None, 1)[1].split(",")
        src_v = set(_regs(ops_[1], "v") + _regs(ops_[2], "v")) if len(ops_) >= 3 else set()
        src_a = set(_regs(ops_[0], "a")) | (set(_regs(ops_[3], "a")) if len(ops_) >= 4 else set())
        scan_up(i - 1, 0, src_v, src_a, t, set())

    # (2) the drain
    last = asm_mfma[-1][0]
    drain = next((j for j in range(last, len(insts)) if insts[j][0].startswith("s_nop 15") and insts[j][1]), None)
    if drain is None:
        findings.append(f"{name}: asm MFMAs but no drain (s_nop 15) behind the last one")
    return findings, len(asm_mfma)


# kernels that MUST show up with asm MFMAs (substring of the mangled name), per source: a build in which one of them has none fails
EXPECTED = {"edge_conv16.hip": ("edge_bwd16_kernel",), "node_mlp16.hip": ("node_mlp_bwd16_kernel",)}


def lint_assembly(path, expected=()):
    """-> (number of kernels scanned, number of kernels with asm MFMAs, [findings]) for one gfx950 assembly file.  ``expected``: kernel
    names (substrings of the mangled names) that must be among the kernels with asm MFMAs -- a missing one is a finding."""
    bad, n, asm_names = [], 0, []
    for name, lines in kernels(path):
        n += 1
        f, n_mfma = lint_kernel_ex(name, lines)
        if n_mfma:
            asm_names.append(name)
        bad += f
    for want in expected:
        if not any(want in k for k in asm_names):
            bad.append(f"{want}: expected a kernel with asm MFMAs of that name in {path}, found {asm_names or 'none'}")
    return n, len(asm_names), bad
