"""CPU-side checks: the C-ABI library loads and exports every symbol include/grl_hip.h declares; host logic without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "grl_hip.h")).read()
    return sorted(set(re.findall(r"\bint\s+(grl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from geometry_rl_amd import hip
    path = hip.build(verbose=False)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/grl_hip.h but not exported by libgrl_hip.so"
    # ... and nothing else: the link uses a version script generated from the header (hip.build), so cross-file helpers
    # (grl_edge16_launch, grl_node_mlp_bwd16_launch, ...) and extern "C" kernels stay internal
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    assert exported == names, (sorted(set(exported) - set(names)), sorted(set(names) - set(exported)))
    # host-only queries are callable without a GPU
    lib.grl_edge_partial_size.restype = ctypes.c_int
    assert lib.grl_edge_partial_size() == 64 * 14 + 64 + 4096 + 64 + 4096
    assert lib.grl_node_mlp_partial_size() == 256 * 64 + 256 + 64 * 256 + 64 * 3
    assert lib.grl_fiber_partial_size() == 16 * 16 * 64 + 64


def test_no_oracle_import_in_product():
    """The product package must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "geometry_rl_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # only inside the CPU-baseline / parity-gate leg (behind the timed region): every oracle import of bench.py sits in that one function
    a = bench.index("def cpu_baseline_and_parity(")
    b = bench.index("\ndef ", a + 1)
    assert bench.count("from oracle") == bench[a:b].count("from oracle") >= 1 and "import oracle" not in bench


def test_product_fails_loudly_without_extension(monkeypatch, tmp_path):
    from geometry_rl_amd import hip
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "missing.so"))
    monkeypatch.setattr(hip, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        hip.lib()


def test_module_state_dict_names_match_reference_layout():
    """SURVEY Appendix B names (incl. PyG tuple-key mangling) so reference checkpoints load."""
    from geometry_rl_amd import agent, graph
    spec = graph.rigid_spec()
    cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device="cpu")
    sd = actor.state_dict()
    for k in ["gnn.ori_grid", "gnn.basis_fn.1.weight", "gnn.basis_fn.3.bias", "gnn.fiber_basis_fn.1.weight", "gnn.node_encoder.weight",
              "gnn.processor.0.convs.<object_geometry___internal___object_geometry>.kernel.weight",
              "gnn.processor.1.convs.<object_geometry___task___grippers>.node_mlp.3.bias",
              "gnn.processor.1.convs.<grippers___agent___grippers>.callibrated", "gnn.decoder.weight", "_pre_std.weight", "_mean.bias"]:
        assert k in sd, k
    n_train = sum(p.numel() for p in actor.parameters())
    assert n_train == 135440
    assert sum(p.numel() for p in critic.parameters()) == 13825
    csd = critic.state_dict()
    for k in ["_network1.gnn.mlp_inner.lins.0.weight", "_network1.gnn.mlp_inner.norms.0.weight", "_network1.gnn.mlp_outer.lins.1.bias",
              "_network1.final.weight"]:
        assert k in csd, k
    assert tuple(sd["gnn.node_encoder.weight"].shape) == (64, 7)


def test_synthetic_shapes_follow_reference_layout():
    from geometry_rl_amd import graph, synthetic as syn
    spec = graph.rigid_spec()
    obs = syn.make_rigid_obs(5)
    for g in spec.in_features:
        base = g.replace("norm_", "")
        assert obs[g].shape == (5, sum(spec.obs_dims[base])), g
    assert obs["position_vectors"].shape[1] == 195 and obs["velocity_vectors"].shape[1] == 12
    c = syn.make_cloth_obs(3)
    assert c["position_vectors"].shape[1] == 1395 and c["velocity_vectors"].shape[1] == 687
    r = syn.make_rope_obs(3)
    assert r["position_vectors"].shape[1] == 486 and r["velocity_vectors"].shape[1] == 246


def test_knockout_switches_need_grl_diag_and_the_product_library_is_not_a_diag_build(tmp_path):
    """the timing knock-outs (wrong results) compile only with -DGRL_DIAG, a GRL_DIAG object exports
    ``grl_diag_build``, and the product library neither exports it nor would be loaded if it did."""
    import subprocess
    from geometry_rl_amd import hip
    src = tmp_path / "t.hip"
    src.write_text('#include "grl_common.h"\n')
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-I", hip.CSRC, str(src)]
    for flag in ("-DGRL_E16_NOGELU", "-DGRL_KNOCK_MFMA", "-DGRL_B16_NOGATHER", "-DGRL_MLPB_NOBARRIER", "-DGRL_FENCED_2W=false"):
        r = subprocess.run(base + [flag], capture_output=True, text=True)
        assert r.returncode != 0 and "GRL_DIAG" in r.stderr, flag
    assert subprocess.run(base + ["-DGRL_E16_NOGELU", "-DGRL_DIAG"], capture_output=True).returncode == 0
    assert subprocess.run(base + ["-DGRL_FENCED_2W=true"], capture_output=True).returncode == 0
    lib = ctypes.CDLL(hip.build(verbose=False))
    assert not hasattr(lib, "grl_diag_build")
    info = __import__("json").load(open(os.path.join(ROOT, "BUILD_INFO.json")))
    assert info["abi_version"] == hip.ABI_VERSION and "build_mode" in info and isinstance(info["objects_rebuilt"], list)


def test_library_is_tied_to_its_sources_by_hash_not_mtime(tmp_path):
    """a tree whose modification times say nothing (all equal, as after a copy to another box) with ONE edited
    source must refuse its stale library until it is rebuilt -- and the rebuild recompiles exactly the object whose source changed."""
    import json
    import shutil
    from geometry_rl_amd import hip
    hip.build(verbose=False)
    root = tmp_path / "tree"
    pkg = root / "geometry_rl_amd"
    shutil.copytree(os.path.join(ROOT, "geometry_rl_amd", "csrc"), pkg / "csrc")           # (objects + their recorded hashes included)
    shutil.copy(os.path.join(ROOT, "geometry_rl_amd", "isa_lint.py"), pkg / "isa_lint.py")
    shutil.copy(os.path.join(ROOT, "geometry_rl_amd", "libgrl_hip.so"), pkg / "libgrl_hip.so")
    os.makedirs(root / "include")
    shutil.copy(os.path.join(ROOT, "include", "grl_hip.h"), root / "include" / "grl_hip.h")
    lib = str(pkg / "libgrl_hip.so")
    h0 = hip.check_library(lib, root=str(root))
    assert h0 == hip.source_hash() == hip.embedded_hash(lib) and len(h0) == 16
    buf = ctypes.create_string_buffer(32)
    assert ctypes.CDLL(lib).grl_source_hash(buf, 32) == 16 and buf.value.decode() == h0
    with open(pkg / "csrc" / "calib.hip", "a") as f:
        f.write("\n// edited\n")
    for d, _, files in os.walk(root):                      # every modification time equal: mtimes must not matter
        for fn in files:
            os.utime(os.path.join(d, fn), (1_700_000_000, 1_700_000_000))
    assert hip.source_hash(str(root)) != h0
    with pytest.raises(RuntimeError, match="built from other sources"):
        hip.check_library(lib, root=str(root))
    hip.build(verbose=False, root=str(root))
    info = json.load(open(root / "BUILD_INFO.json"))
    assert info["objects_rebuilt"] == ["calib.hip.o"], info["objects_rebuilt"]
    assert info["source_hash"] == info["library_source_hash"] == hip.check_library(lib, root=str(root)) != h0
    hip.build(verbose=False, root=str(root))               # and now nothing is left to do
    assert json.load(open(root / "BUILD_INFO.json"))["objects_rebuilt"] == []


def test_balanced_node_order_is_a_reproducible_permutation_with_even_slots():
    """graph.balanced_node_order (round 6): the renumbering the actor's compact graph uses so that the contiguous node ranges of the fused edge
    backward's wave slots carry equal numbers of edges.  kNN-like out-degrees (0 .. 9, mean 3): the natural order cut at node boundaries
    leaves the fullest slot ~25 % over the mean at 24 edges per slot; the windowed longest-processing-time packing stays within 2 edges."""
    from geometry_rl_amd import graph
    g = torch.Generator().manual_seed(4)
    n, slots = 16384, 1024
    deg = torch.poisson(torch.full((n,), 1.5), generator=g).long().clamp(max=9)
    deg[torch.rand(n, generator=g) < 0.5] = 0                     # half of the nodes have no out-edge (padded-away neighbours)
    src = torch.repeat_interleave(torch.arange(n), deg)
    E = int(src.numel())
    new_of_old, split = graph.balanced_node_order(src, n, slots)
    again, split2 = graph.balanced_node_order(src, n, slots)
    assert torch.equal(new_of_old, again) and torch.equal(split, split2)                         # a pure function of the topology
    assert sorted(new_of_old.tolist()) == list(range(n))                                          # a permutation
    assert split.numel() == slots + 1 and int(split[0]) == 0 and int(split[-1]) == n and bool((split[1:] >= split[:-1]).all())
    rp = torch.zeros(n + 1, dtype=torch.long)
    rp[1:] = torch.cumsum(torch.bincount(new_of_old[src], minlength=n), 0)
    load = rp[split.long()[1:]] - rp[split.long()[:-1]]
    assert int(load.sum()) == E and int(load.max()) <= E / slots + 2.5, (int(load.max()), E / slots)
    # the natural order, cut the same way (prefix sums), for comparison: visibly worse
    rp0 = torch.zeros(n + 1, dtype=torch.long)
    rp0[1:] = torch.cumsum(deg, 0)
    cut = torch.searchsorted(rp0.double(), torch.arange(slots + 1, dtype=torch.float64) * (E / slots)).clamp_(max=n)
    cut[0], cut[-1] = 0, n
    assert int((rp0[cut[1:]] - rp0[cut[:-1]]).max()) > int(load.max())
    assert int((new_of_old - torch.arange(n)).abs().max()) < 16 * n // slots * 8                  # nodes stay inside their window


def test_collector_is_held_while_a_stream_captures():
    """agent._no_gc_while_capturing: garbage is collected before the capture and the cyclic collector is disabled for its duration (an object
    with device resources freed inside a capture aborts the process); the collector's previous state comes back, also after an exception."""
    import gc
    from geometry_rl_amd import agent

    class Cycle:
        def __init__(self):
            self.me = self
    seen = []
    import weakref
    c = Cycle()
    ref = weakref.ref(c, lambda _: seen.append("freed"))
    del c
    assert gc.isenabled()
    with agent._no_gc_while_capturing():
        assert seen == ["freed"]            # collected on entry ...
        assert not gc.isenabled()           # ... and nothing is collected inside
    assert gc.isenabled()
    try:
        with agent._no_gc_while_capturing():
            raise KeyError("x")
    except KeyError:
        pass
    assert gc.isenabled()
    gc.disable()
    try:
        with agent._no_gc_while_capturing():
            pass
        assert not gc.isenabled()           # (a caller that runs without the collector keeps it off)
    finally:
        gc.enable()
