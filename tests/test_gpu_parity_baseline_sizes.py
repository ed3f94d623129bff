"""Oracle parity of one whole policy update AT THE SIZES BASELINE.json's configs name (the toy-size twins live in test_gpu_step.py):

  * config 2 exactly: rigid_insertion_multi_hepi_trpl, B = 1024 frames, P = 32 padded object points;
  * config 3's graphs:  cloth_hanging_multi_hepi_trpl with 225 particles / 10 hole points / 4 grippers, B = 256;
  * config 5's graphs:  rope_shaping_hepi_trpl with 80 links / 2 grippers (dim = 2), B = 128;
  * config 4's shard:   rigid two-agent EMPN (2 layers), B = 512 frames = one rank's share of 4096 frames over 8 GPUs.

Checked against the fp32 CPU oracle on identical inputs and parameters: loc, var, state_value, all 13 loss-dict entries
(<= 1e-4 * max(1, |ref|)), every parameter gradient of actor and critic (<= 2e-4 of the tensor's OWN largest reference entry) and the
parameters after the two Adam steps (<= what that gradient tolerance implies through Adam's first step; both rules: tests/parity_util.py).  The same case is also run at a toy batch and BOTH error tables are printed and written to
``gpurun_out/parity_sizes.json``: the products of the MFMA kernels are split-bf16 (three bf16 MFMAs per fp32 product, ~2^-16
relative per product) and the weight gradients sum over millions of rows, so the growth of that error with the length of the
reduction is put on record.  A second oracle run in fp64 attributes the error: ``err(HIP, f64)`` next to ``err(f32 CPU, f64)``.
"""
import json
import os
import time

import numpy as np
import pytest
import torch

from oracle import graph as ogr, step as ost
from geometry_rl_amd import synthetic as syn
from parity_util import G_TOL, adam_first_step_bound, adam_first_step_bound_elem, grad_scales, param_excess

pytestmark = pytest.mark.gpu
LOSS_KEYS = ["loss_objective", "loss_trust_region", "loss_entropy", "loss_critic", "ESS", "kl", "constraint", "mean_constraint",
             "mean_constraint_max", "cov_constraint", "cov_constraint_max", "entropy", "entropy_diff"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # name: (batch at BASELINE size, toy batch)
    "rigid_hepi_b1024": (1024, 16),
    "cloth_hepi_225p": (256, 8),
    "rope_hepi_80l": (128, 8),
    "empn_g2_b512": (512, 16),
    # round 6 (VERDICT r5 "weak" 1): the size the metric is quoted on -- 4096-frame minibatches, the whole update against the oracle
    # (~20 GB and ~1 min of host work per case for the fp32 + fp64 oracle runs: the weight gradients sum over 4 M edge rows)
    "rigid_hepi_b4096": (4096, 16),
    "cloth_hepi_225p_b4096": (4096, 8),
    "empn_g2_b4096": (4096, 16),
}


def _case(name, B):
    from geometry_rl_amd import graph
    if name.startswith("rigid_hepi"):
        o_spec, spec = ogr.rigid_spec(), graph.rigid_spec()
        kw = dict(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)   # configs/rigid_insertion_multi_hepi_trpl_cfg.yaml:110-116
        obs = syn.make_rigid_obs(B, seed=3)
    elif name.startswith("cloth_hepi"):
        o_spec, spec = ogr.cloth_spec(), graph.cloth_spec()
        kw = dict(trust_region_coeff=4.0, cov_bound=0.001)                       # configs/cloth_hanging_multi_hepi_trpl_cfg.yaml:130-133
        obs = syn.make_cloth_obs(B, seed=5)
    elif name.startswith("rope_hepi"):
        o_spec, spec = ogr.rope_spec(), graph.rope_spec()
        kw = dict(dim=2, clip_grad_norm=True)                                    # configs/rope_shaping_hepi_trpl_cfg.yaml
        obs = syn.make_rope_obs(B, seed=6)
    else:
        o_spec = ogr.rigid_spec(G=2, angular_velocity=False, object_velocity=False)
        spec = graph.rigid_spec(G=2, angular_velocity=False, object_velocity=False)
        kw = dict(model="empn")                                                  # configs/rigid_insertion_two_agents_multi_empn_trpl_cfg.yaml
        obs = syn.make_rigid_obs(B, G=2, angular_velocity=False, object_velocity=False, seed=8)
    return o_spec, spec, kw, obs


def _err(got, ref):
    got, ref = torch.as_tensor(got).detach().cpu().double(), torch.as_tensor(ref).detach().cpu().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if not ref.numel():
        return 0.0, 1.0
    return (got - ref).abs().max().item(), max(1.0, ref.abs().max().item())


def _run(name, B, with_f64):
    """One update on both sides -> {tensor name: (abs err, scale, tol)} (+ the f64 attribution table)."""
    from geometry_rl_amd import agent
    from test_gpu_step import load_params
    dev = torch.device("cuda:0")
    o_spec, spec, kw, obs = _case(name, B)
    o_cfg, cfg = ost.AgentConfig(**kw), agent.AgentConfig(**kw)
    a_par, c_par = ost.init_agent_params(o_spec, o_cfg, seed=11)
    oracle = ost.OracleAgent(o_spec, o_cfg, a_par, c_par)
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
    load_params(actor, a_par, dev)
    load_params(critic, {"_network1." + k: v for k, v in c_par.items()}, dev)
    A = spec.num_actuators * cfg.output_dim_vec * 3
    batch = dict(obs)
    batch.update(syn.make_ppo_fields(B, A, seed=B))
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    obs_d = [dbatch[k] for k in spec.in_features]
    table = {}

    def rec(key, got, ref, tol, scale=None):
        e, s = _err(got, ref)
        table[key] = (e, s if scale is None else scale, tol)

    t0 = time.time()
    with torch.no_grad():   # first training call: data-dependent calibration (conv.py:104-105,151-157)
        oracle.actor_forward({k: batch[k] for k in o_spec.in_features}, calibrate=True)
        actor.forward_diag(*obs_d, train=True)
    for k, v in oracle.actor.items():
        if "kernel.weight" in k:
            rec("calibrated " + k, actor.state_dict()[k], v, 1e-4)
    # both sides continue from the oracle-calibrated weights: the update below is compared on its own
    actor.load_state_dict({k: v.detach().to(dev) for k, v in oracle.actor.items()}, strict=False)
    actor._calib_checked = True
    oracle64 = None
    if with_f64:
        oracle64 = ost.OracleAgent(o_spec, o_cfg, {k: v.detach() for k, v in oracle.actor.items()},
                                   {k: v.detach() for k, v in oracle.critic.items()}, dtype=torch.float64)
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, clip_grad_norm=cfg.clip_grad_norm, max_grad_norm=cfg.max_grad_norm)
    ref, ref_grads = oracle.update(batch)
    upd.gflat.zero_()
    out = loss(dbatch)
    (out["loss_objective"] + out["loss_entropy"] + out["loss_trust_region"]).backward()
    out["loss_critic"].backward()
    rec("loc", out["loc"], ref["loc"], 1e-4)
    rec("var", out["sigma"] ** 2, ref["var"], 1e-4)
    rec("state_value", out["state_value"], ref["state_value"], 1e-4)
    for k in LOSS_KEYS:
        rec(k, out[k], ref[k], 1e-4)
    hip_grads = {"actor": {k: p.grad.detach().cpu().clone() for k, p in actor.named_parameters() if k in ref_grads["actor"]},
                 "critic": {k[len("_network1."):]: p.grad.detach().cpu().clone() for k, p in critic.named_parameters()}}
    scales = {net: grad_scales(ref_grads[net]) for net in ("actor", "critic")}
    for net in ("actor", "critic"):
        for k, g in hip_grads[net].items():
            rec(f"grad {net} {k}", g, ref_grads[net][k], G_TOL, scale=scales[net][k])   # against the tensor's OWN scale
    upd.step(dbatch)   # the real step (Adam) from the same starting point
    for net, mod, ref_p, strip in (("actor", actor, oracle.actor, 0), ("critic", critic, oracle.critic, len("_network1."))):
        for k, p in mod.named_parameters():
            kk = k[strip:]
            if cfg.clip_grad_norm or kk not in ref_grads[net]:   # per tensor (gradient clipping rescales what Adam sees)
                rec("param " + k, p, ref_p[kk], adam_first_step_bound(cfg.lr, 1e-5, scales[net].get(kk, 0.0), cfg.clip_grad_norm, p_ref=ref_p[kk]), scale=1.0)
            else:   # entry-wise allowance from the reference gradient: recorded as (worst excess, 1, 1) -- allowed = 1
                table["param " + k] = (param_excess(p, ref_p[kk], adam_first_step_bound_elem(cfg.lr, 1e-5, ref_grads[net][kk], scales[net][kk], p_ref=ref_p[kk])), 1.0, 1.0)
    attribution = None
    if oracle64 is not None:
        ref64, g64 = oracle64.update({k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()})
        attribution = {}
        for key, hip_v, f32_v, r64 in [("loc", out["loc"], ref["loc"], ref64["loc"]),
                                       ("state_value", out["state_value"], ref["state_value"], ref64["state_value"])]:
            attribution[key] = {"hip_vs_f64": _err(hip_v, r64)[0], "cpu_f32_vs_f64": _err(f32_v, r64)[0], "scale": _err(hip_v, r64)[1]}
        for net in ("actor", "critic"):
            worst = {"hip_vs_f64": 0.0, "cpu_f32_vs_f64": 0.0}
            net_max = max((float(v.abs().max()) for v in g64[net].values() if v.numel()), default=0.0)
            for k, g in hip_grads[net].items():
                s = max(1e-30, g64[net][k].abs().max().item(), 1e-4 * net_max)   # (the gradient rule's floor, parity_util.NET_FLOOR)
                worst["hip_vs_f64"] = max(worst["hip_vs_f64"], _err(g, g64[net][k])[0] / s)
                worst["cpu_f32_vs_f64"] = max(worst["cpu_f32_vs_f64"], _err(ref_grads[net][k], g64[net][k])[0] / s)
            attribution[f"grad {net} (worst tensor, relative to its max|g|)"] = worst
    return table, attribution, time.time() - t0


def _summary(table):
    groups = {}
    for k, (e, s, tol) in table.items():
        g = k.split(" ")[0] if k.split(" ")[0] in ("grad", "param", "calibrated") else "values"
        w = groups.setdefault(g, {"worst_err_over_tol": 0.0, "worst": None})
        ratio = e / (tol * s)
        if ratio >= w["worst_err_over_tol"]:
            w.update(worst_err_over_tol=ratio, worst=k, err=e, scale=s, tol=tol)
    return groups


@pytest.mark.parametrize("name", list(CASES))
def test_update_matches_oracle_at_baseline_size(name):
    B_full, B_toy = CASES[name]
    toy, _, _ = _run(name, B_toy, with_f64=False)
    full, attribution, secs = _run(name, B_full, with_f64=True)
    print(f"\n== {name}: B = {B_full} (toy twin B = {B_toy}), oracle + HIP update in {secs:.1f} s")
    print(f"{'tensor':72s} {'err@toy':>10s} {'err@full':>10s} {'allowed':>10s}")
    for k, (e, s, tol) in full.items():
        et = toy.get(k, (float('nan'),))[0]
        print(f"{k:72s} {et:10.2e} {e:10.2e} {tol * s:10.2e}")
    print("attribution against the fp64 oracle:", json.dumps(attribution, indent=1))
    rec = {"batch": B_full, "toy_batch": B_toy, "full": _summary(full), "toy": _summary(toy), "attribution_f64": attribution,
           "seconds": secs}
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "parity_sizes.json")
    allrec = json.load(open(path)) if os.path.exists(path) else {}
    allrec[name] = rec
    json.dump(allrec, open(path, "w"), indent=1)
    bad = {k: (e, tol * s) for k, (e, s, tol) in full.items() if not (np.isfinite(e) and e <= tol * s)}
    assert not bad, bad
    # the fp64 attribution is ASSERTED: against the fp64 oracle every gradient tensor of the HIP path stays within
    # 1e-4 of that tensor's own largest entry (measured 4-7e-5 for the actor's split-bf16 products, ~1e-6 for the critic's fp32 FMAs;
    # the CPU fp32 oracle itself: 2.5-9e-6), and the outputs within 1e-4 of their scale
    for net in ("actor", "critic"):
        w = attribution[f"grad {net} (worst tensor, relative to its max|g|)"]
        assert np.isfinite(w["hip_vs_f64"]) and w["hip_vs_f64"] <= 1e-4, (net, w)
    for key in ("loc", "state_value"):
        a = attribution[key]
        assert a["hip_vs_f64"] <= 1e-4 * max(1.0, a["scale"]), (key, a)
