"""The CPU oracle has the same SO(2) symmetry as the reference model it restates (ponita.py:372-445 demo): grid-preserving rotations
of the inputs rotate the actions and leave the standard deviations alone (dim = 2, 16 orientations: multiples of 22.5 degrees)."""
import math

import pytest
import torch

from geometry_rl_amd import synthetic as syn
from oracle import graph as ogr, step as ost


def _rot(t, c, s):
    v = t.reshape(t.shape[0], -1, 3)
    return torch.stack([c * v[..., 0] - s * v[..., 1], s * v[..., 0] + c * v[..., 1], v[..., 2]], dim=-1).reshape(t.shape)


@pytest.mark.parametrize("k", [3, 8])
def test_oracle_rope_actor_is_rotation_equivariant(k):
    torch.set_num_threads(4)
    spec = ogr.rope_spec(n_links=7, G=2)
    cfg = ost.AgentConfig(dim=2)
    a, c = ost.init_agent_params(spec, cfg, seed=4)
    ag = ost.OracleAgent(spec, cfg, a, c)
    obs = syn.make_rope_obs(4, n_links=7, G=2, seed=9)
    ang = 2.0 * math.pi * k / 16.0
    cs, sn = math.cos(ang), math.sin(ang)
    rot = {k_: (_rot(v, cs, sn) if "vectors" in k_ else v) for k_, v in obs.items()}
    with torch.no_grad():
        ag.actor_forward({k_: obs[k_] for k_ in spec.in_features}, calibrate=True)
        loc, var = ag.actor_forward({k_: obs[k_] for k_ in spec.in_features})
        loc_r, var_r = ag.actor_forward({k_: rot[k_] for k_ in spec.in_features})
    scale = float(loc.abs().max())
    assert scale > 1e-4
    assert float((loc_r - _rot(loc, cs, sn)).abs().max()) <= 1e-5 * scale + 1e-7
    assert float((var_r - var).abs().max()) <= 1e-6 * float(var.abs().max())
