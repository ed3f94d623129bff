"""Synthetic rollouts of the reference observation shapes.

Isaac Sim is out of scope, so the policy-update path is fed with seeded random tensors laid out exactly like the
reference's observation groups (``scalars, position_vectors, velocity_vectors, norm_position_vectors,
norm_velocity_vectors, infos``; geometry_rl/orbit/tasks/manipulation/rigid_tasks/config/common_cfg/
observations_cfg.py:143-192) plus the PPO bookkeeping tensors the loss consumes (examples/torchrl/train.py:249-279).
Pure torch, generated on CPU with a fixed generator so CPU oracle and GPU path see bit-identical inputs.
"""
import math
from typing import Dict

import torch

# valid object points per env (env % 8); mesh USD blobs are absent from the reference checkout, so these stand in
# for the per-shape point counts.
RIGID_NUM_POINTS = (24, 10, 20, 16, 8, 32, 12, 6)


def _clip(x, lim):
    return x.clamp_(-lim, lim)


def make_rigid_obs(B: int, *, P: int = 32, G: int = 1, E_mesh: int = 180, angular_velocity: bool = True,
                   object_velocity: bool = True, seed: int = 0, env_offset: int = 0) -> Dict[str, torch.Tensor]:
    """One batch of rigid-task observations.  Padded object/target points are zero in the raw position tensors
    (geometry_rl/orbit/tasks/common/utils.py:193-211); clipping follows configs/rigid_insertion_multi_hepi_trpl_cfg.yaml:69-72."""
    g = torch.Generator().manual_seed(seed)
    env = torch.arange(B) + env_offset
    num_points = torch.tensor(RIGID_NUM_POINTS)[env % 8].clamp(max=P)
    valid = (torch.arange(P)[None, :] < num_points[:, None]).float()[..., None]  # [B,P,1]
    grip = torch.rand(B, G, 3, generator=g) * 2 - 1
    obj = (torch.rand(B, P, 3, generator=g) * 2 - 1) * valid
    tgt = (obj + 0.3 * (torch.rand(B, 1, 3, generator=g) * 2 - 1)) * valid
    pos = torch.cat([grip.reshape(B, -1), obj.reshape(B, -1), tgt.reshape(B, -1)], dim=1)
    n_vel = G * (2 if angular_velocity else 1) + (2 if angular_velocity else 1) * (1 if object_velocity else 0)
    vel = _clip(torch.randn(B, 3 * n_vel, generator=g), 20.0)
    obs = {
        "scalars": _clip(torch.randn(B, 1, generator=g), 20.0),
        "position_vectors": _clip(pos, 20.0),
        "velocity_vectors": vel,
        "norm_position_vectors": _clip(torch.randn(B, pos.shape[1], generator=g), 20.0),
        "norm_velocity_vectors": _clip(torch.randn(B, vel.shape[1], generator=g), 20.0),
    }
    infos = torch.zeros(B, 1 + 2 * E_mesh + 1)
    infos[:, 0] = num_points.float()
    infos[:, -1] = float(E_mesh)
    obs["infos"] = infos
    return obs


def make_cloth_obs(B: int, *, n_particles: int = 225, n_hole: int = 10, G: int = 4, E_cloth: int = 600,
                   seed: int = 0) -> Dict[str, torch.Tensor]:
    """cloth_tasks/config/common_cfg/observations_cfg.py:150-193 layout; clip per configs/cloth_hanging_multi_hepi_trpl_cfg.yaml."""
    g = torch.Generator().manual_seed(seed)
    n_pos = G + 2 * n_particles + n_hole + 1
    n_vel = G + n_particles
    return {
        "scalars": _clip(torch.randn(B, n_hole + E_cloth, generator=g), 50.0),
        "position_vectors": torch.rand(B, 3 * n_pos, generator=g) * 2 - 1,
        "velocity_vectors": _clip(torch.randn(B, 3 * n_vel, generator=g), 50.0),
        "norm_position_vectors": _clip(torch.randn(B, 3 * n_pos, generator=g), 50.0),
        "norm_velocity_vectors": _clip(torch.randn(B, 3 * n_vel, generator=g), 50.0),
    }


ROPE_NUM_LINKS = (80, 40)   # valid links per env (env % 2) of the variable-length synthetic ropes: shaping 80 / closing 40 links


def make_rope_obs(B: int, *, n_links: int = 80, G: int = 2, seed: int = 0, variable_length: bool = False,
                  env_offset: int = 0) -> Dict[str, torch.Tensor]:
    """rope_tasks/config/common_cfg/observations_cfg.py:131-160 layout.  ``variable_length``: adds ``infos`` = links_num_points [B,1]
    (ROPE_NUM_LINKS by env % 2, scaled by n_links / 80) and zero-pads the raw link / target positions beyond it."""
    obs = _make_rope_obs(B, n_links=n_links, G=G, seed=seed)
    if variable_length:
        env = torch.arange(B) + env_offset
        nl = torch.tensor([max(2, v * n_links // 80) for v in ROPE_NUM_LINKS])[env % 2]   # 80 / 40 links at full size
        valid = (torch.arange(n_links)[None, :] < nl[:, None]).float()[..., None]       # [B,L,1]
        pos = obs["position_vectors"].clone()
        for blk in range(2):   # links, target_geometry
            sl = slice(3 * G + blk * 3 * n_links, 3 * G + (blk + 1) * 3 * n_links)
            pos[:, sl] = (pos[:, sl].reshape(B, n_links, 3) * valid).reshape(B, -1)
        obs["position_vectors"] = pos
        obs["infos"] = nl.float().reshape(B, 1)
    return obs


def _make_rope_obs(B: int, *, n_links: int = 80, G: int = 2, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    n_pos = G + 2 * n_links
    n_vel = G + n_links
    return {
        "scalars": _clip(torch.randn(B, 1, generator=g), 10.0),
        "position_vectors": torch.rand(B, 3 * n_pos, generator=g) * 2 - 1,
        "velocity_vectors": _clip(torch.randn(B, 3 * n_vel, generator=g), 10.0),
        "norm_position_vectors": _clip(torch.randn(B, 3 * n_pos, generator=g), 10.0),
        "norm_velocity_vectors": _clip(torch.randn(B, 3 * n_vel, generator=g), 10.0),
    }


def make_ppo_fields(B: int, A: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Per-frame tensors the TRPL loss reads (examples/torchrl/train.py:249-279; trpl.py:231-253,176-229):
    action, old loc / diagonal covariance, sample_log_prob (consistent with them), old state_value, advantage,
    value_target."""
    g = torch.Generator().manual_seed(seed + 7919)
    loc = torch.randn(B, A, generator=g)
    var = torch.rand(B, A, generator=g) + 0.5
    action = loc + var.sqrt() * torch.randn(B, A, generator=g)
    logp = -0.5 * (((action - loc) ** 2 / var).sum(-1) + A * math.log(2 * math.pi) + var.log().sum(-1))
    value = torch.randn(B, 1, generator=g)
    adv = torch.randn(B, 1, generator=g)
    return {"action": action, "loc": loc, "var": var, "sample_log_prob": logp, "state_value": value,
            "advantage": adv, "value_target": adv + value}


def make_gae_inputs(N: int, T: int, seed: int = 0, episode_len: int = 100) -> Dict[str, torch.Tensor]:
    """reward ~ N(0,1); done every ``episode_len`` steps (rigid_insertion_multi_env_cfg.py:290), terminated = 0;
    critic values [N, T+1] stand in for the shifted critic pass (train.py:134-140)."""
    g = torch.Generator().manual_seed(seed + 104729)
    t = torch.arange(T)
    done = ((t % episode_len) == episode_len - 1)[None, :].expand(N, T).contiguous()
    return {"reward": torch.randn(N, T, generator=g), "done": done, "terminated": torch.zeros(N, T, dtype=torch.bool),
            "values": torch.randn(N, T + 1, generator=g)}
