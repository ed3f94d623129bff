"""ctypes binding of the C-ABI kernel library ``libgrl_hip.so`` (declared in include/grl_hip.h).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved the import of this module (and
therefore every op of the package) raises.  Tensors are passed as raw device pointers + the current HIP stream.
"""
import ctypes
import os
import subprocess
import sys
from typing import Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("GRL_LIB", os.path.join(_HERE, "libgrl_hip.so"))  # GRL_LIB: debugging builds only
ABI_VERSION = 205   # include/grl_hip.h GRL_HIP_VERSION
SOURCES = ["edge_conv.hip", "edge_conv16.hip", "node_ops.hip", "node_mlp.hip", "node_mlp16.hip", "head_ops.hip",
           "critic_ops.hip", "train_ops.hip", "weight_images.hip", "calib.hip", "oneshot.hip"]
# (source, extra flags, object suffix): the two MFMA files are compiled a second time as the plain-bf16 variant (one MFMA per
# product instead of three; csrc/grl_common.h GRL_PREC) whose entry points carry the suffix _bf16
# per-source compiler flags.  edge_conv16.hip: its 512-register backward kernel keeps the chain's MFMA results in VGPRs (the default
# selection would put every MFMA result of such a kernel into AGPRs) and pins the weight-gradient tiles to AGPRs itself (asm)
# node_mlp16.hip additionally without SLP vectorisation: beside the MFMAs of its one wave per SIMD plain f32 instructions overlap with the matrix
# pipe, packed ones (v_pk_*) do not (DESIGN.md finding 23)
# (edge_conv16.hip the same, with the scalar GELU in its one-wave backward: +1 % on the step, profiles/r03_ab_edge16_noslp.txt)
FILE_FLAGS = {"edge_conv16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize", "-DGRL_B16_SCALAR_GELU"],
              "node_mlp16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
              # the ConvNeXt forward: plain-f32 GELU as well (0.313 vs 0.330 ms per step; the 16-row edge forward, three waves per SIMD, is
              # FASTER with its packed GELU: 0.48 vs 0.52 ms -- profiles/r03_ab_forward_gelu_ld1.txt)
              "node_mlp.hip": ["-fno-slp-vectorize", "-DGRL_GELU4_SCALAR=1"]}
VARIANTS = [("edge_conv.hip", ["-DGRL_PREC=1"], ".bf16"), ("edge_conv16.hip", ["-DGRL_PREC=1"], ".bf16"),
            ("node_mlp.hip", ["-DGRL_PREC=1"], ".bf16"), ("node_mlp16.hip", ["-DGRL_PREC=1"], ".bf16"),
            ("node_ops.hip", ["-DGRL_PREC=1"], ".bf16"), ("weight_images.hip", ["-DGRL_PREC=1"], ".bf16")]


BUILD_INFO = os.path.join(os.path.dirname(_HERE), "BUILD_INFO.json")
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
HASH_TAG = b"GRL_SOURCE_HASH="


def _sha(*chunks) -> str:
    import hashlib
    h = hashlib.sha256()
    for c in chunks:
        h.update(c if isinstance(c, bytes) else str(c).encode())
        h.update(b"\0")
    return h.hexdigest()


def _read(path) -> bytes:
    with open(path, "rb") as f:
        return f.read()


def _pkg_paths(root=None):
    """(csrc directory, include/grl_hip.h, isa_lint.py) of the package tree at ``root`` (default: this checkout)."""
    pkg = os.path.join(root, "geometry_rl_amd") if root else _HERE
    return os.path.join(pkg, "csrc"), os.path.join(os.path.dirname(pkg), "include", "grl_hip.h"), os.path.join(pkg, "isa_lint.py")


def source_hash(root=None) -> str:
    """16 hex digits over EVERYTHING libgrl_hip.so is made from: every .hip / .h under csrc/, include/grl_hip.h (the export list), the
    ISA lint's rules and the flag tables of this module.  ``hip.build()`` embeds it in the library (``grl_source_hash``), ``hip.lib()``
    recomputes it from the tree and refuses a library that was built from other sources -- modification times play no part (a tree copied
    to another box has arbitrary ones: VERDICT r5)."""
    import json
    csrc, header, lint = _pkg_paths(root)
    names = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    parts = [json.dumps([BASE_FLAGS, SOURCES, FILE_FLAGS, VARIANTS], sort_keys=True)]
    for n in names:
        parts += [n, _read(os.path.join(csrc, n))]
    parts += ["grl_hip.h", _read(header), "isa_lint.py", _read(lint)]
    return _sha(*parts)[:16]


def embedded_hash(path) -> str:
    """The source hash a built library carries (read from the file's bytes: no dlopen), or "" for a library without one."""
    import re
    try:
        m = re.search(HASH_TAG + rb"([0-9a-f]{16})", _read(path))
    except OSError:
        return ""
    return m.group(1).decode() if m else ""


def check_library(path=None, root=None) -> str:
    """Raise unless the library at ``path`` was built from the sources of the tree at ``root``; -> the hash."""
    path = path or LIB_PATH
    want, have = source_hash(root), embedded_hash(path)
    if want != have:
        raise RuntimeError(f"{path} was built from other sources than this tree (library {have or 'carries no source hash'}, tree {want}): "
                           "rebuild the extension (python -c 'import __graft_entry__ as g; g.build()')")
    return want


def build(verbose: bool = True, force: bool = False, root=None) -> str:
    """Compile every HIP source for gfx950 and link libgrl_hip.so in-tree (hipcc cross-compiles without a GPU).  What is rebuilt is decided
    by CONTENT: an object is kept when the hash recorded beside it (its source, every header, its flags, the lint rules) still matches, the
    library when its embedded ``grl_source_hash`` equals ``source_hash()`` of the tree.  Writes BUILD_INFO.json at the repository root
    ("prebuilt" = nothing compiled).  ``root``: build another checkout of this repository (tests)."""
    import json
    import time
    all_flags = [f for fl in FILE_FLAGS.values() for f in fl] + [f for _, fl, _ in VARIANTS for f in fl]
    if any(f.startswith("-DGRL_DIAG") for f in all_flags):
        raise RuntimeError("GRL_DIAG (timing knock-outs: wrong results) must never be built into libgrl_hip.so -- use tools/build_variants.sh")
    csrc, header, lint_py = _pkg_paths(root)
    lib_path = os.path.join(os.path.dirname(csrc), "libgrl_hip.so") if root else LIB_PATH
    info_path = os.path.join(root, "BUILD_INFO.json") if root else BUILD_INFO
    srcs = [os.path.join(csrc, s) for s in SOURCES if os.path.exists(os.path.join(csrc, s))]
    tree_hash = source_hash(root)

    def info(mode, rebuilt, lint=None):
        try:
            with open(info_path, "w") as f:
                json.dump({"build_mode": mode, "objects_rebuilt": rebuilt, "isa_lint": lint, "objects_total": len(srcs) + len(VARIANTS) + 1,
                           "library": os.path.relpath(lib_path, os.path.dirname(os.path.dirname(csrc))),
                           "library_bytes": os.path.getsize(lib_path) if os.path.exists(lib_path) else None, "abi_version": ABI_VERSION,
                           "source_hash": tree_hash, "library_source_hash": embedded_hash(lib_path),
                           "forced": bool(force), "time": time.strftime("%Y-%m-%dT%H:%M:%S")}, f, indent=1)
        except OSError:
            pass
    if not force and embedded_hash(lib_path) == tree_hash:
        info("prebuilt (the library's embedded source hash equals the tree's: nothing compiled)", [])
        return lib_path
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    from . import isa_lint
    header_bytes = [_read(os.path.join(csrc, h)) for h in sorted(f for f in os.listdir(csrc) if f.endswith(".h"))]   # every header feeds every object
    lint_bytes = _read(lint_py)
    objs = []
    procs = []
    rebuilt = []
    lint_jobs = []   # (label, assembly file, object, its hash file, its hash) of every object compiled NOW from a source with asm MFMAs
    pending = []     # (hash file, hash) of the other objects compiled now: recorded once hipcc has succeeded
    bdir = os.path.join(csrc, "build")
    os.makedirs(bdir, exist_ok=True)
    jobs = [(s, [], "") for s in srcs] + [(os.path.join(csrc, s), fl, sfx) for s, fl, sfx in VARIANTS]
    for s, flags, sfx in jobs:
        base = os.path.basename(s)
        lint = base in isa_lint.FILES
        # sources with asm MFMAs are compiled with -save-temps into a directory of their own: the assembly the REAL compile leaves behind
        # is what the ISA lint reads (a different hipcc or different flags must not get past the hazard check)
        odir = os.path.join(bdir, base + sfx + ".d") if lint else bdir
        os.makedirs(odir, exist_ok=True)
        o = os.path.join(odir, base + sfx + ".o")
        objs.append(o)
        cmd_flags = BASE_FLAGS + FILE_FLAGS.get(base, []) + flags + (["-save-temps=obj"] if lint else [])
        ohash = _sha(_read(s), *header_bytes, " ".join(cmd_flags), lint_bytes if lint else b"")
        hfile = o + ".srchash"
        if not force and os.path.exists(o) and os.path.exists(hfile) and _read(hfile).decode().strip() == ohash:
            continue
        for stale in (o, hfile):   # never leave an object behind whose recorded hash could vouch for it after a failed compile
            if os.path.exists(stale):
                os.remove(stale)
        cmd = [hipcc] + cmd_flags + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        rebuilt.append(os.path.basename(o))
        if lint:
            lint_jobs.append((base + sfx, os.path.join(odir, os.path.splitext(base)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s"), o, hfile, ohash, base))
        else:
            pending.append((hfile, ohash))
        procs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE if not verbose else None)))
    failed = []
    for cmd, p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            failed.append(" ".join(cmd) + ("\n" + err.decode(errors="replace")[-4000:] if err else ""))
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    for hfile, ohash in pending:
        with open(hfile, "w") as f:
            f.write(ohash)
    # lint EVERY job before judging any: an object's hash file is written only when its assembly passed, and every object that failed is
    # removed -- a later incremental build can neither link an unchecked object nor skip its lint
    lint_report, lint_failed = {}, []
    for label, asm, o, hfile, ohash, base in lint_jobs:
        n, n_asm, bad = isa_lint.lint_assembly(asm, isa_lint.EXPECTED.get(base, ()))
        lint_report[label] = {"kernels": n, "kernels_with_asm_mfma": n_asm, "findings": bad}
        if bad or n_asm == 0:
            if os.path.exists(o):
                os.remove(o)
            lint_failed.append(f"{label}: " + ("; ".join(bad) if bad else "no kernel with asm MFMAs found in the assembly -- the lint saw nothing"))
        else:
            with open(hfile, "w") as f:
                f.write(ohash)
    if lint_failed:
        raise RuntimeError("ISA lint failed (geometry_rl_amd/isa_lint.py): " + " | ".join(lint_failed))
    # exports = exactly the entry points include/grl_hip.h declares (cross-file helpers such as grl_edge16_launch stay internal)
    import re
    names = sorted(set(re.findall(r"\bint\s+(grl_[a-z0-9_]+)\s*\(", open(header).read())))
    vs = os.path.join(bdir, "exports.map")
    with open(vs, "w") as f:
        f.write("{\n  global:\n" + "".join(f"    {n};\n" for n in names) + "  local: *;\n};\n")
    # the tree's source hash, embedded: grl_source_hash() hands it out, embedded_hash() finds it in the file's bytes
    hsrc = os.path.join(bdir, "source_hash.cpp")
    with open(hsrc, "w") as f:
        f.write('// generated by geometry_rl_amd/hip.py build()\n'
                f'static const char grl_hash_tag[] = "{HASH_TAG.decode()}{tree_hash}";\n'
                'extern "C" int grl_source_hash(char* buf, int cap) {\n'
                '  const char* h = grl_hash_tag + %d; int n = 0;\n'
                '  while (h[n] && n + 1 < cap) { buf[n] = h[n]; ++n; }\n'
                '  if (cap > 0) buf[n] = 0;\n  return n;\n}\n' % len(HASH_TAG))
    hobj = os.path.join(bdir, "source_hash.o")
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O1", "-fPIC", "-c", hsrc, "-o", hobj])
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vs, "-o", lib_path] + objs + [hobj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    if embedded_hash(lib_path) != tree_hash:
        raise RuntimeError(f"{lib_path}: the embedded source hash did not survive the link")
    info("compiled from source" if len(rebuilt) == len(objs) else "incremental (objects whose recorded source hash still matched were kept)", rebuilt, lint_report)
    return lib_path


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension must be built (python -c 'import __graft_entry__ as g; g.build()'). "
                "geometry_rl_amd has no CPU or PyTorch fallback path.")
        _lib = ctypes.CDLL(LIB_PATH)
        if hasattr(_lib, "grl_diag_build") and not os.environ.get("GRL_ALLOW_DIAG_LIB"):
            raise RuntimeError(f"{LIB_PATH} is a GRL_DIAG build (timing knock-outs: its results are wrong).  Diagnostic libraries are loaded "
                               "only with GRL_ALLOW_DIAG_LIB=1 (tools/run_variants.sh); rebuild the product library with __graft_entry__.build()")
        # the binary must be the one THESE sources build (an explicitly named debugging library -- GRL_LIB, tools/r0*_ab_libs.sh -- is the
        # caller's business; GRL_ALLOW_STALE_LIB=1 switches the check off for bisecting)
        if "GRL_LIB" not in os.environ and os.environ.get("GRL_ALLOW_STALE_LIB", "0") == "0":
            check_library(LIB_PATH)
        _lib.grl_version.restype = ctypes.c_int
        if _lib.grl_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports ABI version {_lib.grl_version()}, this package binds version {ABI_VERSION} "
                               "(include/grl_hip.h GRL_HIP_VERSION): rebuild the extension (__graft_entry__.build())")
    return _lib


def _arg(a):
    if torch.is_tensor(a):
        if not a.is_contiguous():
            raise ValueError("non-contiguous tensor passed to a HIP kernel")
        return ctypes.c_void_p(a.data_ptr())
    if a is None:
        return ctypes.c_void_p(0)
    if isinstance(a, bool):
        return ctypes.c_int(int(a))
    if isinstance(a, int):
        return ctypes.c_int(a)
    if isinstance(a, float):
        return ctypes.c_float(a)
    return a


def stream_ptr() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-entry-point timing with HIP events recorded on the launch stream (bench.py's roofline leg).
# KERNEL_TIMES maps name -> list of (start_event, end_event); enable with ``hip.KERNEL_TIMES = {}``.
KERNEL_TIMES = None
KERNEL_ROWS = {}   # name -> rows (edge x orientation / node x orientation) handed to the entry point while timing is on


def call(name: str, *args, stream=None, rows: int = 0) -> None:
    """Launch C-ABI entry point ``name`` on the current stream; raises on a non-zero status."""
    fn = getattr(lib(), name)
    fn.restype = ctypes.c_int
    timing = KERNEL_TIMES is not None
    if timing:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = fn(*[_arg(a) for a in args], stream if stream is not None else stream_ptr())
    if timing:
        e1.record()
        KERNEL_TIMES.setdefault(name, []).append((e0, e1))
        KERNEL_ROWS[name] = KERNEL_ROWS.get(name, 0) + rows
    if rc != 0:
        raise RuntimeError(f"{name} failed with status {rc}")


def kernel_time_summary():
    """-> {name: (n_calls, total_ms)} from the recorded events (synchronises)."""
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in (KERNEL_TIMES or {}).items()}


def kernel_prof_enable(on: bool) -> None:
    """In-library HIP-event timing of the kernels inside multi-kernel entry points (grl_prof_*)."""
    lib().grl_prof_enable(ctypes.c_int(int(on)))


def kernel_prof_summary():
    """-> {kernel name: (n_launches, total_ms)} from the in-library records (synchronises on each record)."""
    l = lib()
    out = {}
    buf = ctypes.create_string_buffer(96)
    ms = ctypes.c_float(0.0)
    for i in range(l.grl_prof_count()):
        if l.grl_prof_get(ctypes.c_int(i), buf, ctypes.c_int(96), ctypes.byref(ms)) != 0:
            raise RuntimeError("grl_prof_get failed")
        n, t = out.get(buf.value.decode(), (0, 0.0))
        out[buf.value.decode()] = (n + 1, t + ms.value)
    return out


def query(name: str, *args) -> int:
    """Call a pure host-side query entry point (no stream argument)."""
    fn = getattr(lib(), name)
    fn.restype = ctypes.c_int
    return fn(*[_arg(a) for a in args])


def check_f32(*tensors: torch.Tensor):
    for t in tensors:
        if t is not None and (t.dtype != torch.float32 or not t.is_cuda):
            raise TypeError(f"expected a CUDA float32 tensor, got {t.dtype} on {t.device}")


def storage_dtype(prec: str) -> torch.dtype:
    """Storage type of the node latents for an entry-point suffix: "" -> float32, "_bf16" -> bfloat16 (csrc/grl_common.h st_t)."""
    return torch.bfloat16 if prec == "_bf16" else torch.float32


def check_latent(prec: str, *tensors: torch.Tensor):
    want = storage_dtype(prec)
    for t in tensors:
        if t is not None and (t.dtype != want or not t.is_cuda):
            raise TypeError(f"expected a CUDA {want} latent tensor for the '{prec or 'fp32'}' kernels, got {t.dtype} on {t.device}")


_hiprt = None


def stream_wait_value32(value_tensor: torch.Tensor, value: int) -> None:
    """The CURRENT stream waits (in the command processor: no wave, no compute unit is held) until the int32 at ``value_tensor`` is >=
    ``value`` (hipStreamWaitValue32, flag hipStreamWaitValueGte).  The location must be written by work already enqueued on another
    stream -- the caller owns that ordering."""
    global _hiprt
    if _hiprt is None:
        _hiprt = ctypes.CDLL("libamdhip64.so")
        _hiprt.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
        _hiprt.hipStreamWaitValue32.restype = ctypes.c_int
    rc = _hiprt.hipStreamWaitValue32(stream_ptr(), ctypes.c_void_p(value_tensor.data_ptr()), ctypes.c_uint32(int(value) & 0xFFFFFFFF), 0, 0xFFFFFFFF)
    if rc != 0:
        raise RuntimeError(f"hipStreamWaitValue32 failed with status {rc}")
