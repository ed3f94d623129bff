"""SO(2) equivariance of the HIP actor (property tests modelled on the reference's demo, ponita.py:372-445): with
``ponita_dim=2`` the 16 orientations are the 16th roots of unity, so a rotation of every input vector about z by a multiple of
22.5 degrees only permutes the fiber -- the action vectors must rotate with the inputs, the standard deviations must not move.
(The S2 Fibonacci grid of the 3-D tasks has no such discrete symmetry; there equivariance is approximate by construction.)"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rot(t, c, s):
    v = t.reshape(t.shape[0], -1, 3)
    return torch.stack([c * v[..., 0] - s * v[..., 1], s * v[..., 0] + c * v[..., 1], v[..., 2]], dim=-1).reshape(t.shape)


@pytest.mark.parametrize("k", [1, 4, 11])
def test_rope_actor_is_rotation_equivariant(k):
    from geometry_rl_amd import agent, graph, synthetic as syn
    dev = torch.device("cuda:0")
    spec = graph.rope_spec(n_links=9, G=2)
    cfg = agent.AgentConfig(dim=2)
    torch.manual_seed(2)
    actor, _, _, _ = agent.build_agent(spec, cfg, device=dev)
    B = 5
    obs = {k_: v.to(dev) for k_, v in syn.make_rope_obs(B, n_links=9, G=2, seed=7).items()}
    ang = 2.0 * math.pi * k / 16.0
    c, s = math.cos(ang), math.sin(ang)
    rot = {k_: (_rot(v, c, s) if "vectors" in k_ else v) for k_, v in obs.items()}
    with torch.no_grad():
        actor.forward_diag(*[obs[k_] for k_ in spec.in_features], train=True)   # calibration on the un-rotated batch
        loc, sig = actor.forward_diag(*[obs[k_] for k_ in spec.in_features], train=False)
        loc_r, sig_r = actor.forward_diag(*[rot[k_] for k_ in spec.in_features], train=False)
    assert float(loc.abs().max()) > 1e-4
    scale = float(loc.abs().max())
    print(f"k={k}: equivariance error {float((loc_r - _rot(loc, c, s)).abs().max()) / scale:.2e} (relative), sigma {float((sig_r - sig).abs().max()):.2e}")
    assert float((loc_r - _rot(loc, c, s)).abs().max()) <= 2e-4 * scale + 1e-6
    assert float((sig_r - sig).abs().max()) <= 1e-5 * float(sig.abs().max())
    # and a rotation that is NOT on the grid is not a symmetry of the discretised fiber: the check above is not vacuous
    off = 2.0 * math.pi * 0.37 / 16.0
    rot2 = {k_: (_rot(v, math.cos(off), math.sin(off)) if "vectors" in k_ else v) for k_, v in obs.items()}
    with torch.no_grad():
        loc_o, _ = actor.forward_diag(*[rot2[k_] for k_ in spec.in_features], train=False)
    assert float((loc_o - _rot(loc, math.cos(off), math.sin(off))).abs().max()) > 1e-3 * scale
