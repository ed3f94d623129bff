"""Wiring of actor, critic, projection and loss for one task config -- the counterpart of
``examples/torchrl/builders/agent.py:14-80`` + ``builders/utils_algo_graph.py:208-276`` without Hydra / the Isaac env object
(observation layout comes from a TaskSpec instead of ``env.observation_manager``), plus the policy-update driver that
replaces the inner loop of ``examples/torchrl/train.py:258-316``."""
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import hip, ops
from .graph import HyperData, TaskSpec
from .hepi import HEPi, FiberBundleConv
from .policy import BaseCritic, DeepSets, GNNGaussianPolicyDiag, GNNVFNet
from .trpl import KLProjectionLayer, TRPLLoss


@dataclass
class AgentConfig:
    """Values of configs/<task>_hepi_trpl_cfg.yaml that reach the hot path."""
    model: str = "hepi"
    dim: int = 3
    num_ori: int = 16
    only_upper_hemisphere: bool = False
    output_dim: int = 1
    output_dim_vec: int = 1
    num_layers: int = 2
    codes: tuple = ((1, 0), (0, 1), (0, 1))  # configs/algorithm/pyg_agent/model/hepi.yaml:17-48
    init_std: float = 1.0
    minimal_std: float = 1e-5
    mean_bound: float = 0.05
    cov_bound: float = 0.0025
    trust_region_coeff: float = 1.0
    entropy_coef: float = 0.005
    critic_coef: float = 0.5
    clip_value: float = 0.2
    lr: float = 3e-4
    clip_grad_norm: bool = False
    max_grad_norm: float = 1.0


def build_agent(spec: TaskSpec, cfg: AgentConfig, device="cuda", group=None):
    """-> (actor GNNGaussianPolicyDiag, critic BaseCritic, projection, loss_module)  (agent.py:31-52)."""
    n_in = len(spec.node_types) + spec.n_vec  # utils_algo_graph.py:79
    if cfg.model == "hepi":
        mp = []  # utils_algo_graph.py:29-47: one fresh conv per (level, active round)
        for lvl in range(len(spec.edge_levels)):
            mp.append([FiberBundleConv(64, 64, 64, groups=64, separable=True, widening_factor=4) if cfg.codes[lvl][k] else None
                       for k in range(len(cfg.codes[lvl]))])
        gnn = HEPi(input_dim_node=n_in, input_dim_edge=len(spec.edge_types) + 4, hidden_dim=64, latent_dim=64,
                   output_dim=cfg.output_dim, output_dim_vec=cfg.output_dim_vec, node_type_mapping=spec.node_types,
                   edge_type_mapping=[tuple(e) for e in spec.edge_types], edge_level_mapping=spec.edge_levels,
                   message_passing=mp, num_messages=len(cfg.codes[0]), device=device, num_ori=cfg.num_ori,
                   ponita_dim=cfg.dim, only_upper_hemisphere=cfg.only_upper_hemisphere)
    elif cfg.model == "empn":
        from .ponita_gcn import PonitaGCN
        gnn = PonitaGCN(input_dim_node=n_in, output_dim=cfg.output_dim, output_dim_vec=cfg.output_dim_vec,
                        num_layers=cfg.num_layers, hidden_dim=64, num_ori=cfg.num_ori, ponita_dim=cfg.dim,
                        only_upper_hemisphere=cfg.only_upper_hemisphere, device=device)
    else:
        raise ValueError(cfg.model)
    a_data = HyperData(spec, full_graph_obs=False, dist_as_pos=True, output_mask_key=spec.actuator, concat_input_vector=False)
    A = spec.num_actuators * cfg.output_dim_vec * 3
    actor = GNNGaussianPolicyDiag(gnn=gnn, hyper_data=a_data, action_dim=A, num_actuators=spec.num_actuators, init="orthogonal",
                                  hidden_sizes=(64, 64), contextual_std=True, init_std=cfg.init_std, minimal_std=cfg.minimal_std,
                                  share_action_dim=True, post_fc=False)
    c_data = HyperData(spec, full_graph_obs=True, dist_as_pos=False, output_mask_key=None, concat_input_vector=True)
    c_gnn = DeepSets(input_dim_node=len(spec.node_types) + 3 * spec.n_vec, output_dim=64, hidden_dim=64, device=device)
    critic = BaseCritic(GNNVFNet(gnn=c_gnn, hyper_data=c_data))
    critic._network1.group = group
    projection = KLProjectionLayer(proj_type="kl", mean_bound=cfg.mean_bound, cov_bound=cfg.cov_bound,
                                   trust_region_coeff=cfg.trust_region_coeff, scale_prec=True, entropy_schedule=False, action_dim=A)
    loss = TRPLLoss(actor, critic, projection=projection, entropy_coef=cfg.entropy_coef, critic_coef=cfg.critic_coef,
                    clip_value=cfg.clip_value, loss_critic_type="l2", normalize_advantage=True, in_features=spec.in_features,
                    group=group)
    return actor, critic, projection, loss


class PolicyUpdater:
    """One policy-update step = loss forward, actor + critic backward, optional clip_grad_norm_ per network, two Adam(lr,
    eps=1e-5) steps (train.py:279-316).  Parameters of both networks live in ONE flat fp32 buffer (gradients likewise), so a
    data-parallel run needs a single RCCL all-reduce per step and Adam is a single kernel."""

    def __init__(self, loss_module: TRPLLoss, lr=3e-4, eps=1e-5, betas=(0.9, 0.999), clip_grad_norm=False, max_grad_norm=1.0,
                 group=None):
        self.loss_module, self.group = loss_module, group
        self.lr, self.eps, self.betas = lr, eps, betas
        self.clip, self.max_norm = clip_grad_norm, max_grad_norm
        a = [p for p in loss_module.actor_network.parameters() if p.requires_grad]
        c = [p for p in loss_module.critic_network.parameters() if p.requires_grad]
        self.params = a + c
        self.n_actor = sum(p.numel() for p in a)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.gflat = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.gflat[off:off + k].view_as(p)
            off += k
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.steps = 0
        if group is not None:  # replicas start identical (parameter init incl. calibration is rank 0's)
            import torch.distributed as dist
            dist.broadcast(self.flat, src=dist.get_global_rank(group, 0) if hasattr(dist, "get_global_rank") else 0, group=group)

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        self.loss_module._global_steps = self.steps
        self.gflat.zero_()
        out = self.loss_module(batch)
        actor_loss = out["loss_objective"] + out["loss_entropy"] + out["loss_trust_region"]  # train.py:296-301
        torch.autograd.backward([actor_loss, out["loss_critic"]])  # train.py:304-305
        if self.group is not None:
            import torch.distributed as dist
            dist.all_reduce(self.gflat, group=self.group)  # loss terms are already scaled by 1/B_global
        self.steps += 1
        na, n = self.n_actor, self.flat.numel()
        segs = [(0, na), (na, n)]
        for lo, hi in segs:
            coef = None
            if self.clip:  # train.py:308-310
                sq = torch.zeros(1, device=self.flat.device, dtype=torch.float64)
                coef = torch.empty(1, device=self.flat.device, dtype=torch.float32)
                hip.call("grl_clip_coef", self.gflat[lo:hi], hi - lo, float(self.max_norm), sq, coef)
            hip.call("grl_adam_step", self.flat[lo:hi], self.gflat[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], hi - lo,
                     float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), self.steps, coef, 1.0)
        return out


def gae(reward, done, terminated, values, gamma=0.99, lmbda=0.95):
    """Shifted GAE (train.py:134-140,249-251): reward/done/terminated [N,T], values [N,T+1] -> advantage, value_target [N,T]."""
    hip.check_f32(reward, values)
    N, T = reward.shape
    adv = torch.empty_like(reward)
    tgt = torch.empty_like(reward)
    hip.call("grl_gae_scan", reward.contiguous(), done.to(torch.uint8).contiguous(), terminated.to(torch.uint8).contiguous(),
             values.contiguous(), adv, tgt, N, T, float(gamma), float(lmbda))
    return adv, tgt
