"""The committed fixture of one full policy update at a tiny shape (tools/make_step_fixture.py): the oracle must keep
reproducing it (CPU), the HIP path must hit it without the oracle in the loop (GPU): every loss-dict entry, the pre-clip gradients
and the post-Adam parameters (clip_grad_norm on, so the clip coefficient is covered too)."""
import os

import numpy as np
import pytest
import torch

LOSS_KEYS = ("loss_objective", "loss_trust_region", "loss_entropy", "loss_critic", "ESS", "kl", "constraint", "mean_constraint",
             "mean_constraint_max", "cov_constraint", "cov_constraint_max", "entropy", "entropy_diff")


def _load(golden_dir):
    z = np.load(os.path.join(golden_dir, "step_rigid_g2_tiny.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _sub(z, prefix):
    return {k[len(prefix):]: v for k, v in z.items() if k.startswith(prefix)}


def _close(name, a, b, tol):
    a, b = a.detach().cpu().double(), b.double()
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    err = float((a - b).abs().max()) if b.numel() else 0.0
    assert err <= tol * scale, (name, err, scale)


def test_oracle_reproduces_the_step_fixture(golden_dir):
    from oracle import graph as ogr, step as ost
    z = _load(golden_dir)
    spec = ogr.rigid_spec(P=8, G=2, E_mesh=4, angular_velocity=False, object_velocity=False)
    cfg = ost.AgentConfig(clip_grad_norm=True, max_grad_norm=0.5)
    ag = ost.OracleAgent(spec, cfg, {k: v.clone() for k, v in _sub(z, "actor0.").items()}, {k: v.clone() for k, v in _sub(z, "critic0.").items()})
    out, grads = ag.update(_sub(z, "in."))
    for k in LOSS_KEYS:
        _close(k, out[k], z["out." + k], 1e-6)
    for k, v in _sub(z, "grad.actor.").items():
        _close("grad " + k, grads["actor"][k], v, 1e-6)
    for k, v in _sub(z, "actor1.").items():
        if v.dtype.is_floating_point:
            _close("param " + k, ag.actor[k], v, 1e-6)
    for k, v in _sub(z, "critic1.").items():
        _close("param " + k, ag.critic[k], v, 1e-6)


@pytest.mark.gpu
def test_hip_update_hits_the_step_fixture(golden_dir):
    from geometry_rl_amd import agent, graph
    dev = torch.device("cuda:0")
    z = _load(golden_dir)
    spec = graph.rigid_spec(P=8, G=2, E_mesh=4, angular_velocity=False, object_velocity=False)
    cfg = agent.AgentConfig(clip_grad_norm=True, max_grad_norm=0.5)
    actor, critic, _, loss = agent.build_agent(spec, cfg, device=dev)
    actor.load_state_dict({k: v.to(dev) for k, v in _sub(z, "actor0.").items()}, strict=False)   # calibrated weights, flags set
    critic.load_state_dict({"_network1." + k: v.to(dev) for k, v in _sub(z, "critic0.").items()}, strict=True)
    actor._calib_checked = True
    batch = {k: v.to(dev) for k, v in _sub(z, "in.").items()}
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, clip_grad_norm=True, max_grad_norm=0.5)
    # gradients before clipping / Adam: the loss module alone
    out = loss(batch)
    (out["loss_objective"] + out["loss_entropy"] + out["loss_trust_region"]).backward()
    out["loss_critic"].backward()
    for k in LOSS_KEYS:
        _close(k, out[k], z["out." + k], 1e-4)
    _close("loc", out["loc"], z["out.loc"], 1e-4)
    for k, p in actor.named_parameters():
        if "grad.actor." + k in z:
            _close("grad " + k, p.grad, z["grad.actor." + k], 2e-4)
    for k, p in critic.named_parameters():
        _close("grad " + k, p.grad, z["grad.critic." + k[len("_network1."):]], 2e-4)
    upd.gflat.zero_()
    upd.step(batch)
    sd = actor.state_dict()
    for k, v in _sub(z, "actor1.").items():
        if v.dtype.is_floating_point and k in sd:
            _close("param " + k, sd[k], v, 5e-5)
    sdc = critic.state_dict()
    for k, v in _sub(z, "critic1.").items():
        _close("param " + k, sdc["_network1." + k], v, 5e-5)
