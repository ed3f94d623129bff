"""K = 5 CONSECUTIVE policy updates against the CPU oracle.

One Adam step from zero moments moves every parameter by ~lr * sign(g): it pins signs and the Adam kernel, not gradient magnitudes.  From
the second step on the moments are non-zero, the update depends on the RATIO of accumulated gradients, and an error in a gradient's
magnitude shows up in exp_avg, exp_avg_sq and the parameters.  Both sides run five updates on five different minibatches from identical
(oracle-calibrated) parameters; after the fifth:

  * exp_avg      of every parameter tensor within  5e-4 of the tensor's own largest reference entry
  * exp_avg_sq   of every parameter tensor within  1e-3 of the tensor's own largest reference entry   (squares: twice the relative error)
    (floors as for gradients, tests/parity_util.py: a tensor's scale is at least 1e-4 of the largest entry of its network)
  * parameters   within  K * (the first-step bound of parity_util.adam_first_step_bound) -- the trajectories are compared, not re-synchronised:
                 the bound is what a gradient error of 2e-4 per step can accumulate to through Adam

Cases: rigid HEPi B = 64, cloth HEPi B = 16 (25 particles), two-agent EMPN B = 32 -- and, since round 5, the sizes
BASELINE.json names: rigid HEPi at config 2's exact minibatch (B = 1024, P = 32, K = 5: ~15 s of oracle time on the GPU box) and cloth with
225 particles / 10 hole points / 4 grippers at B = 256 (K = 3).  The recorded (hipGraph) step is what runs from the third update on, so the
comparison also covers replayed launches.  Reference semantics: examples/torchrl/train.py:264-316."""
import os

import numpy as np
import pytest
import torch

from oracle import step as ost
from geometry_rl_amd import synthetic as syn
from parity_util import NET_FLOOR, adam_first_step_bound, grad_scales

pytestmark = pytest.mark.gpu
M_TOL, V_TOL = 5e-4, 1e-3


def _obs(name, B, seed):
    if name == "rigid_g1":
        return syn.make_rigid_obs(B, seed=seed)
    if name == "cloth":
        return syn.make_cloth_obs(B, n_particles=25, E_cloth=40, seed=seed)
    if name == "cloth_full":   # 225 particles, 10 hole points, 4 grippers
        return syn.make_cloth_obs(B, seed=seed)
    return syn.make_rigid_obs(B, G=2, angular_velocity=False, object_velocity=False, seed=seed)


@pytest.mark.parametrize("name,B,K", [("rigid_g1", 64, 5), ("cloth", 16, 5), ("empn_g2", 32, 5), ("rigid_g1", 1024, 5), ("cloth_full", 256, 3)])
def test_five_updates_match_the_oracle(name, B, K):
    from geometry_rl_amd import agent, graph
    from oracle import graph as ogr
    from test_gpu_step import load_params, make_case
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(32, os.cpu_count() or 1))   # the oracle: more intra-op threads than this only slow its small CPU ops down
    if name == "cloth_full":
        o_spec, spec, kw = ogr.cloth_spec(), graph.cloth_spec(), dict(trust_region_coeff=4.0, cov_bound=0.001)
    else:
        o_spec, spec, kw, _ = make_case(name, B)
    o_cfg, cfg = ost.AgentConfig(**kw), agent.AgentConfig(**kw)
    a_par, c_par = ost.init_agent_params(o_spec, o_cfg, seed=21)
    oracle = ost.OracleAgent(o_spec, o_cfg, a_par, c_par)
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
    load_params(actor, a_par, dev)
    load_params(critic, {"_network1." + k: v for k, v in c_par.items()}, dev)
    A = spec.num_actuators * cfg.output_dim_vec * 3
    batches = []
    for i in range(K):
        b = dict(_obs(name, B, 30 + i))
        b.update(syn.make_ppo_fields(B, A, seed=40 + i))
        batches.append(b)
    with torch.no_grad():   # first training call: calibration on the first minibatch, then identical weights on both sides
        oracle.actor_forward({k: batches[0][k] for k in o_spec.in_features}, calibrate=True)
    actor.load_state_dict({k: v.detach().to(dev) for k, v in oracle.actor.items()}, strict=False)
    for mod in actor.modules():
        if hasattr(mod, "callibrated"):
            mod.callibrated.fill_(True)
    actor._calib_checked = True
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, clip_grad_norm=cfg.clip_grad_norm, max_grad_norm=cfg.max_grad_norm, use_graph=True)
    g_scale = None
    for i, b in enumerate(batches):
        ref, ref_grads = oracle.update(b)
        out = upd.step({k: v.to(dev) for k, v in b.items()})
        sc = {net: grad_scales(ref_grads[net]) for net in ("actor", "critic")}
        g_scale = sc if g_scale is None else {net: {k: max(v, g_scale[net].get(k, 0.0)) for k, v in sc[net].items()} for net in sc}
        for k in ("loss_objective", "loss_trust_region", "loss_critic", "kl"):   # the trajectories stay together step by step
            e = abs(float(out[k]) - float(ref[k]))
            assert e <= 1e-4 * max(1.0, abs(float(ref[k]))), (i, k, e)
    assert upd.mode.startswith("graph") and upd._program is not None
    torch.cuda.synchronize()
    off = lambda p: (p.data_ptr() - upd.flat.data_ptr()) // 4
    bad, worst = [], {"exp_avg": 0.0, "exp_avg_sq": 0.0, "param": 0.0}
    for net, mod, ref_p, optim, strip in (("actor", actor, oracle.actor, oracle.actor_optim, 0),
                                          ("critic", critic, oracle.critic, oracle.critic_optim, len("_network1."))):
        states = {kk: optim.state.get(ref_p[kk], {}) for kk in ref_p}
        m_ref = {kk: s_["exp_avg"] for kk, s_ in states.items() if "exp_avg" in s_}
        v_ref = {kk: s_["exp_avg_sq"] for kk, s_ in states.items() if "exp_avg_sq" in s_}
        m_sc, v_sc = grad_scales(m_ref), grad_scales(v_ref)
        for k, p in mod.named_parameters():
            kk = k[strip:]
            if kk not in m_ref:
                continue
            o, n = off(p), p.numel()
            em = float((upd.exp_avg[o:o + n].view_as(p).cpu().double() - m_ref[kk].double()).abs().max())
            ev = float((upd.exp_avg_sq[o:o + n].view_as(p).cpu().double() - v_ref[kk].double()).abs().max())
            ep = float((p.detach().cpu().double() - ref_p[kk].detach().double()).abs().max())
            allowed_p = K * adam_first_step_bound(cfg.lr, 1e-5, g_scale[net].get(kk, 0.0), cfg.clip_grad_norm, p_ref=ref_p[kk])
            print(f"{net} {kk}: exp_avg {em / m_sc[kk]:.2e} of scale, exp_avg_sq {ev / v_sc[kk]:.2e} of scale, param err {ep:.2e} (allowed {allowed_p:.2e})")
            worst["exp_avg"] = max(worst["exp_avg"], em / m_sc[kk])
            worst["exp_avg_sq"] = max(worst["exp_avg_sq"], ev / v_sc[kk])
            worst["param"] = max(worst["param"], ep / allowed_p)
            if not (em <= M_TOL * m_sc[kk] and ev <= V_TOL * v_sc[kk] and ep <= allowed_p and np.isfinite(em + ev + ep)):
                bad.append((net, kk, em / m_sc[kk], ev / v_sc[kk], ep, allowed_p))
    print("worst (fraction of scale / of allowed):", worst)
    assert not bad, bad
