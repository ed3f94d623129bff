"""The HIP HEPi against the REFERENCE's own numbers (tests/golden/tier2b_*.npz, written by tools/make_golden.py from the reference
code): the reference ``state_dict`` is loaded by name (checkpoint compatibility, train.py:336-368 / SURVEY 8f.3: PyG ModuleDict key
mangling, ``callibrated`` buffers), then first-call outputs, the calibrated weights, post-calibration outputs and every parameter
gradient are compared at the 1e-4 bar of BASELINE.json -- no oracle in between."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    "rigid_g1": dict(spec=lambda g: g.rigid_spec(P=8, G=1, E_mesh=4), dim=3, upper=True, od=2, ov=2),
    "rigid_g2": dict(spec=lambda g: g.rigid_spec(P=8, G=2, E_mesh=4, angular_velocity=False, object_velocity=False),
                     dim=3, upper=False, od=1, ov=1),
    "rope_dim2": dict(spec=lambda g: g.rope_spec(n_links=7, G=2), dim=2, upper=False, od=1, ov=1),
    # hepi_attention.yaml: FiberBundleConv(aggr="AttentionalAggregation") -- per-edge messages, gate network, per-destination softmax
    "rigid_g2_attention": dict(spec=lambda g: g.rigid_spec(P=8, G=2, E_mesh=4, angular_velocity=False, object_velocity=False),
                               dim=3, upper=False, od=1, ov=1, aggr="AttentionalAggregation"),
}


def close(a, b, atol, rtol=1e-4, what=""):
    a, b = a.detach().cpu().double(), b.double()
    err = (a - b).abs()
    assert bool((err <= atol + rtol * b.abs()).all()), (what, err.max().item(), b.abs().max().item())


@pytest.mark.parametrize("name", list(CASES))
def test_hepi_matches_reference_fixture(golden_dir, name):
    from geometry_rl_amd import agent, graph
    dev = torch.device("cuda:0")
    c = CASES[name]
    spec = c["spec"](graph)
    z = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, f"tier2b_hepi_{name}.npz")).items()}
    cfg = agent.AgentConfig(dim=c["dim"], only_upper_hemisphere=c["upper"], output_dim=c["od"], output_dim_vec=c["ov"],
                            aggr=c.get("aggr", "add"))
    actor, _, _, _ = agent.build_agent(spec, cfg, device=dev)
    gnn = actor.gnn

    # --- reference checkpoint, by name
    init = {k[5:]: v.to(dev) for k, v in z.items() if k.startswith("init.")}
    sd = gnn.state_dict()
    assert set(init) == set(sd), set(init) ^ set(sd)
    for k in sd:
        assert sd[k].shape == init[k].shape, k
    gnn.load_state_dict(init, strict=True)
    assert not gnn.calibrated

    obs = [z["obs." + k].to(dev) for k in spec.in_features]
    g, u = actor.hyper_data.build_data(*obs, train=True)
    for et, es in g.edges.items():   # same edge SET as the reference graph (order inside a destination segment is free)
        ref = z["edge_index." + "|".join(et)]
        # (in the NATURAL node numbering: the package renumbers the main node type for the edge backward's load balance, GraphBatch.natural)
        mine = torch.stack([g.natural(et[0], es.src_d).cpu().long(), g.natural(et[2], es.dst_d).cpu().long()])
        key = lambda e: sorted(map(tuple, e.t().tolist()))
        assert key(mine) == key(ref), et
    assert sum(z["edge_index." + "|".join(et)].shape[1] > 0 for et in spec.edge_types) == len(g.edges)

    # --- calibrating call: its outputs come from the un-rescaled activations (conv.py:104-105,151-157)
    with torch.no_grad():
        out0, hid0 = gnn.one_step(g, u)
    close(out0, z["out_first_call"], 1e-4, what="out_first_call")
    close(hid0, z["hidden_first_call"], 1e-4, what="hidden_first_call")
    gnn.calibrate(g, u)
    for k, v in gnn.state_dict().items():
        ref = z["cal." + k]
        if ref.dtype.is_floating_point:
            close(v, ref, 1e-4, what="cal." + k)
        else:
            assert bool(v.cpu()) == bool(ref), k

    # --- post-calibration forward + every parameter gradient
    gnn.load_state_dict({k[4:]: v.to(dev) for k, v in z.items() if k.startswith("cal.")}, strict=True)
    gnn.zero_grad()
    out, hid = gnn.one_step(g, u)
    close(out, z["out"], 1e-4, what="out")
    close(hid, z["hidden"], 1e-4, what="hidden")
    ((out * z["R_out"].to(dev)).sum() + (hid * z["R_hidden"].to(dev)).sum()).backward()
    n = 0
    for k, p in gnn.named_parameters():
        if "grad." + k in z:
            ref = z["grad." + k]
            # attention: d gate = alpha dx1 (msg - x1) cancels leading digits of the split-bf16 messages before the sums over all rows
            # (gradient bar of the update tests: 2e-4 of the tensor's largest entry)
            if c.get("aggr"):
                # The gate network ends in a ReLU: a pre-activation within rounding distance of 0 may take the other branch on the two
                # sides, which moves single entries of the gradients behind it by O(dgate * msg) -- a property of the function, not
                # of the kernels (the same update agrees with the oracle to 4e-5 in tests/test_gpu_step.py[rigid_attn]).  Compared in
                # the Frobenius norm instead of entry by entry.
                a, b = p.grad.detach().cpu().double(), ref.double()
                rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
                print(f"grad.{k}: relative Frobenius error {rel:.2e}, max|err| {float((a - b).abs().max()):.2e}, max|ref| {float(b.abs().max()):.2e}")
                if "gate_nn" in k:
                    # d gate sums to ZERO over every softmax group, so the gate network's gradients are what the ReLU mask leaves of
                    # terms that cancel: ill-conditioned by construction.  Bar: absolute, 5e-4 of the largest gradient entry of the net.
                    gmax = max(float(v.abs().max()) for kk, v in z.items() if kk.startswith("grad."))
                    assert float((a - b).abs().max()) <= 5e-4 * gmax, ("grad." + k, float((a - b).abs().max()), gmax)
                else:
                    assert rel <= 3e-4, ("grad." + k, rel)
            else:
                close(p.grad, ref, 1e-4 * max(1.0, ref.abs().max().item()), what="grad." + k)
            n += 1
    assert n >= 20


def test_reference_checkpoint_round_trip(tmp_path):
    """train.py:336-368 / play.py:194-205 checkpoint layout: save -> load into a fresh agent -> identical policy and value."""
    from geometry_rl_amd import agent, graph, synthetic as syn
    dev = torch.device("cuda:0")
    spec = graph.rigid_spec(P=8, G=1, E_mesh=4)
    cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
    obs = {k: v.to(dev) for k, v in syn.make_rigid_obs(6, P=8, G=1, E_mesh=4, seed=3).items()}
    args = [obs[k] for k in spec.in_features]
    torch.manual_seed(0)
    actor, critic, _, _ = agent.build_agent(spec, cfg, device=dev)
    with torch.no_grad():
        actor(*args, train=True)            # calibrates: the ``callibrated`` buffers travel with the checkpoint
        loc, cov = actor(*args, train=False)
        val = critic(*args)
    path = tmp_path / "model_checkpoint_best.pth"
    torch.save(agent.reference_checkpoint(actor, critic, reward=1.5), path)
    ck = torch.load(path, weights_only=False)
    assert all(k.startswith("module.0.module.") for k in ck["actor"]) and all(k.startswith("module.") for k in ck["critic"])
    assert any("<object_geometry___internal___object_geometry>" in k for k in ck["actor"])
    torch.manual_seed(1)
    actor2, critic2, _, _ = agent.build_agent(spec, cfg, device=dev)
    assert agent.load_reference_checkpoint(path, actor2, critic2) == 1.5
    flags = {et: bool(c.callibrated) for r in actor2.gnn.processor for et, c in r.items()}
    assert flags[("object_geometry", "internal", "object_geometry")] and not flags[("grippers", "agent", "grippers")]  # G=1: no agent edges
    with torch.no_grad():
        loc2, cov2 = actor2(*args, train=True)   # must NOT re-calibrate
        val2 = critic2(*args)
    assert torch.equal(loc, loc2) and torch.equal(cov, cov2) and torch.equal(val, val2)


@pytest.mark.parametrize("name", ["frob", "w2"])
def test_projection_layers_match_reference_fixture(golden_dir, name):
    """Frobenius / Wasserstein projections of the fused kernel against the reference layers' own outputs (tier2c fixtures)."""
    from geometry_rl_amd import trpl
    dev = torch.device("cuda:0")
    z = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, f"tier2c_projection_{name}.npz")).items()}
    layer = (trpl.FrobeniusProjectionLayer if name == "frob" else trpl.WassersteinProjectionLayer)(
        mean_bound=float(z["mean_bound"]), cov_bound=float(z["cov_bound"]), trust_region_coeff=float(z["coeff"]))
    pm, pS = layer(None, (z["mean"].to(dev), z["S"].to(dev)), (z["mean_o"].to(dev), z["S_o"].to(dev)))
    close(pm, z["proj_mean"], 2e-6, what="proj_mean")
    close(pS, z["proj_S"], 2e-6, what="proj_S")
