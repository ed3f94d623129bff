"""Build-time ISA lint of the asm-pinned MFMA accumulators (geometry_rl_amd/isa_lint.py).  ``hip.build()`` runs it on the
assembly of every object it compiles from a file with asm MFMAs and fails the build on a finding; here (CPU only, hipcc cross-compiles):
the rules catch what they claim to catch on hand-made assembly, and a forced rebuild of the two files passes with its lint report written."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MFMA = "\tv_mfma_f32_32x32x16_bf16 a[0:15], v[10:13], v[14:17], a[0:15]"
DRAIN = ["\t;;#ASMSTART", "\ts_nop 15", "\ts_nop 15", "\t;;#ASMEND", "\ts_endpgm"]


def _lint(body):
    from geometry_rl_amd import isa_lint
    lines = body + DRAIN
    return isa_lint.lint_kernel("_Zk", lines)


def _asm(*insts):
    return ["\t;;#ASMSTART", *insts, "\t;;#ASMEND"]


def test_clean_stream_passes():
    assert _lint(["\tv_add_f32 v10, v1, v2", "\ts_nop 1", *_asm(MFMA)]) == []
    assert _lint(["\tv_add_f32 v10, v1, v2", "\tv_mov_b32 v40, v41", "\tv_mov_b32 v42, v41", *_asm(MFMA)]) == []


def test_source_written_one_wait_state_before_is_caught():
    f = _lint(["\tv_add_f32 v10, v1, v2", "\tv_mov_b32 v40, v41", *_asm(MFMA)])
    assert len(f) == 1 and "writes a source" in f[0]


def test_accumulator_tile_written_by_accvgpr_write_is_caught():   # SrcC: the "+a" tile itself
    f = _lint(["\tv_accvgpr_write_b32 a5, v3", *_asm(MFMA)])
    assert len(f) == 1 and "accumulator tile" in f[0]
    assert _lint(["\tv_accvgpr_write_b32 a5, v3", "\ts_nop 1", *_asm(MFMA)]) == []
    assert _lint(["\tv_accvgpr_write_b32 a16, v3", *_asm(MFMA)]) == []   # another tile


def test_second_destination_of_swap_is_caught():
    f = _lint(["\tv_swap_b32 v90, v15", *_asm(MFMA)])
    assert len(f) == 1 and "v_swap_b32" in f[0]


def test_loop_back_edge_predecessor_is_caught():
    # the MFMA is the first instruction of a loop body; its source is written at the tail of the previous iteration, behind s_cbranch
    body = ["\tv_mov_b32 v60, v61", "\tv_mov_b32 v60, v61", ".LBB0_1:", *_asm(MFMA), "\tv_mov_b32 v70, v71", "\tv_mov_b32 v70, v71",
            "\tv_add_f32 v14, v1, v2", "\ts_cbranch_scc1 .LBB0_1"]
    f = _lint(body)
    assert len(f) == 1 and "v_add_f32 v14" in f[0] and "1 wait states" in f[0]
    body[-2], body[-3] = body[-3], body[-2]    # one more instruction between the write and the branch: two wait states
    assert _lint(body) == []


def test_read_of_pinned_tile_inside_the_window_is_caught():
    f = _lint([*_asm(MFMA), "\tv_accvgpr_read_b32 v3, a7"])
    assert len(f) == 1 and "v_accvgpr_read_b32" in f[0]


def test_missing_drain_is_caught():
    from geometry_rl_amd import isa_lint
    f = isa_lint.lint_kernel("_Zk", [*_asm(MFMA), "\ts_endpgm"])
    assert any("no drain" in x for x in f)


def test_build_lints_the_objects_it_compiles(tmp_path):
    """remove the objects of the linted files (and the library): exactly those are recompiled, linted, and the report is written"""
    from geometry_rl_amd import hip, isa_lint
    for f in isa_lint.FILES:
        for sfx in ("", ".bf16"):
            o = os.path.join(hip.CSRC, "build", f + sfx + ".d", f + sfx + ".o")
            if os.path.exists(o):
                os.remove(o)
    lib = hip.LIB_PATH
    if os.path.exists(lib):
        os.remove(lib)          # no library whose embedded hash could say "prebuilt": build() goes through its object loop
    hip.build(verbose=False)
    info = json.load(open(hip.BUILD_INFO))
    rep = info["isa_lint"]
    assert set(rep) == {f + sfx for f in isa_lint.FILES for sfx in ("", ".bf16")}, rep
    assert all(v["kernels_with_asm_mfma"] >= 1 and v["findings"] == [] for v in rep.values()), rep
    assert sorted(info["objects_rebuilt"]) == sorted(f + sfx + ".o" for f in isa_lint.FILES for sfx in ("", ".bf16")), info["objects_rebuilt"]


def test_kernel_without_asm_mfma_is_not_counted_and_expected_names_are_enforced(tmp_path):
    """an ASMSTART block that is only an empty pin statement beside a builtin MFMA is NOT an asm-MFMA kernel; a build in which
    an expected kernel shows no asm MFMA fails."""
    from geometry_rl_amd import isa_lint
    asm = tmp_path / "k.s"
    asm.write_text("\n".join(["_Z17edge_bwd16_kernelv:", "\t;;#ASMSTART", "\t;;#ASMEND", MFMA, "\ts_endpgm", ""]))
    n, n_asm, bad = isa_lint.lint_assembly(str(asm), ("edge_bwd16_kernel",))
    assert (n, n_asm) == (1, 0) and len(bad) == 1 and "expected a kernel with asm MFMAs" in bad[0]
    asm.write_text("\n".join(["_Z17edge_bwd16_kernelv:", *_asm(MFMA), *DRAIN, ""]))
    assert isa_lint.lint_assembly(str(asm), ("edge_bwd16_kernel",)) == (1, 1, [])


def test_read_behind_a_branch_inside_the_window_is_caught():
    body = [*_asm(MFMA), "\ts_cbranch_scc1 .LBB0_2", "\tv_mov_b32 v1, v2", *DRAIN[:-1], "\ts_branch .LBB0_3", ".LBB0_2:", "\tv_accvgpr_read_b32 v3, a7",
            ".LBB0_3:"]
    f = _lint(body)
    assert len(f) == 1 and "v_accvgpr_read_b32" in f[0]


def test_command_line_form_is_clean():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_acc_lint.py")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "accumulator lint: clean" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
