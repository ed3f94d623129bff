"""Oracle: one full policy-update step (actor + critic forward, TRPL loss, two backward passes, two Adam steps).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates examples/torchrl/train.py:258-316 around the oracle
pieces in equivariant.py / graph.py / trpl.py.  Also the timed "cpu_baseline" (kind = "port") of bench.py.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import equivariant as eq
from . import graph as gr
from . import trpl as tr


@dataclass
class AgentConfig:
    model: str = "hepi"  # "hepi" | "empn"
    dim: int = 3
    num_ori: int = 16
    only_upper_hemisphere: bool = False
    output_dim: int = 1
    output_dim_vec: int = 1
    num_layers: int = 2  # empn
    codes: tuple = ((1, 0), (0, 1), (0, 1))  # configs/algorithm/pyg_agent/model/hepi.yaml:17-48
    init_std: float = 1.0
    minimal_std: float = 1e-5
    mean_bound: float = 0.05
    cov_bound: float = 0.0025
    proj_type: str = "kl"  # kl | frob | w2 (configs/algorithm/projection/*.yaml)
    trust_region_coeff: float = 1.0
    entropy_coef: float = 0.005
    critic_coef: float = 0.5
    clip_value: float = 0.2
    lr: float = 3e-4
    clip_grad_norm: bool = False
    max_grad_norm: float = 1.0
    aggr: str = "add"  # "AttentionalAggregation": configs/algorithm/pyg_agent/model/hepi_attention.yaml


class OracleAgent:
    """Functional actor/critic with reference state_dict names; parameters are leaf tensors."""

    def __init__(self, spec: gr.TaskSpec, cfg: AgentConfig, actor_params: Dict[str, torch.Tensor],
                 critic_params: Dict[str, torch.Tensor], dtype=torch.float32):
        self.spec, self.cfg, self.dtype = spec, cfg, dtype
        self.rounds = eq.hepi_schedule(spec.edge_types, spec.edge_levels, [list(c) for c in cfg.codes])
        self.actor = {k: v.detach().clone().to(dtype) for k, v in actor_params.items()}
        self.critic = {k: v.detach().clone().to(dtype) for k, v in critic_params.items()}
        self._topo = {}
        self._trainable(True)
        self.actor_optim = torch.optim.Adam(self._actor_leaves(), lr=cfg.lr, eps=1e-5)  # train.py:145
        self.critic_optim = torch.optim.Adam(list(self.critic.values()), lr=cfg.lr, eps=1e-5)  # train.py:146

    def _is_buffer(self, k):
        return k.endswith("ori_grid")

    def _actor_leaves(self) -> List[torch.Tensor]:
        return [v for k, v in self.actor.items() if not self._is_buffer(k)]

    def _trainable(self, flag):
        for k, v in self.actor.items():
            if not self._is_buffer(k):
                v.requires_grad_(flag)
        for v in self.critic.values():
            v.requires_grad_(flag)

    # -- graph (topology cached per batch size like rigid.py:254-255)
    def _graph(self, obs, full_graph_obs, dist_as_pos):
        obs = {k: v.to(self.dtype) for k, v in obs.items()}
        split = gr.split_obs(self.spec, obs)
        key = (obs["scalars"].shape[0], full_graph_obs)
        if key not in self._topo:
            self._topo[key] = gr.build_topology(self.spec, split, full_graph_obs)
        topo = self._topo[key]
        graph, s, v = gr.build_features(self.spec, topo, split, dist_as_pos)
        graph["output_mask_key"] = "grippers"
        return topo, graph, s, v

    def actor_forward(self, obs, calibrate=False):
        """gnn_gaussian_policy_diag.py:26-87 with post_fc=False, contextual_std=True -> (loc [B,A], var [B,A])."""
        c = self.cfg
        topo, graph, s, v = self._graph(obs, full_graph_obs=False, dist_as_pos=True)
        B = topo["batch_size"]
        gnnP = {k[len("gnn."):]: p for k, p in self.actor.items() if k.startswith("gnn.")}
        if c.model == "hepi":
            out, hidden = eq.hepi_forward(gnnP, graph, s, v, dim=c.dim, output_dim=c.output_dim,
                                          output_dim_vec=c.output_dim_vec, rounds=self.rounds, calibrate=calibrate)
        else:
            out, hidden = eq.empn_forward(gnnP, graph, s, v, dim=c.dim, output_dim=c.output_dim,
                                          output_dim_vec=c.output_dim_vec, num_layers=c.num_layers, batch_size=B,
                                          calibrate=calibrate)
        if calibrate:
            for k, p in gnnP.items():
                if not self._is_buffer(k) and p is not self.actor["gnn." + k]:
                    self.actor["gnn." + k].data.copy_(p)
        std = tr.std_head(hidden, self.actor["_pre_std.weight"], self.actor["_pre_std.bias"], c.init_std,
                          c.minimal_std, B)
        return out.reshape(B, -1), std ** 2  # diag of diag_embed(std)**2

    def critic_forward(self, obs, stats_fn=None):
        topo, graph, s, v = self._graph(obs, full_graph_obs=True, dist_as_pos=False)
        return gr.value_forward(self.critic, gr.critic_input(topo, s, v), stats_fn)

    def loss(self, batch: Dict[str, torch.Tensor], adv_stats=None, stats_fn=None):
        c = self.cfg
        b = {k: (v.to(self.dtype) if v.is_floating_point() else v) for k, v in batch.items()}
        obs = {k: b[k] for k in self.spec.in_features}
        loc, var = self.actor_forward(obs)
        value = self.critic_forward(obs, stats_fn)
        out = tr.trpl_loss(loc, var, b, value, mean_bound=c.mean_bound, cov_bound=c.cov_bound,
                           trust_region_coeff=c.trust_region_coeff, entropy_coef=c.entropy_coef,
                           critic_coef=c.critic_coef, clip_value=c.clip_value, adv_stats=adv_stats, proj_type=c.proj_type)
        out["loc"], out["var"], out["state_value"] = loc, var, value
        return out

    def update(self, batch):
        """train.py:279-316: loss, actor/critic backward, optional clip, Adam x2, zero_grad."""
        out = self.loss(batch)
        actor_loss = out["loss_objective"] + out["loss_entropy"] + out["loss_trust_region"]
        actor_loss.backward()
        out["loss_critic"].backward()
        grads = {"actor": {k: v.grad.clone() for k, v in self.actor.items() if v.grad is not None},
                 "critic": {k: v.grad.clone() for k, v in self.critic.items() if v.grad is not None}}  # before clipping
        if self.cfg.clip_grad_norm:
            torch.nn.utils.clip_grad_norm_(self._actor_leaves(), self.cfg.max_grad_norm)
            torch.nn.utils.clip_grad_norm_(list(self.critic.values()), self.cfg.max_grad_norm)
        self.actor_optim.step()
        self.critic_optim.step()
        self.actor_optim.zero_grad()
        self.critic_optim.zero_grad()
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}, grads


def init_agent_params(spec: gr.TaskSpec, cfg: AgentConfig, seed=0):
    """Random-init parameters with the reference's module structure."""
    n_in = len(spec.node_types) + spec.n_vec
    rounds = eq.hepi_schedule(spec.edge_types, spec.edge_levels, [list(c) for c in cfg.codes])
    if cfg.model == "hepi":
        g = eq.init_hepi_params(n_in, rounds, dim=cfg.dim, num_ori=cfg.num_ori,
                                only_upper_hemisphere=cfg.only_upper_hemisphere, output_dim=cfg.output_dim,
                                output_dim_vec=cfg.output_dim_vec, seed=seed)
    else:
        g = eq.init_empn_params(n_in, dim=cfg.dim, num_ori=cfg.num_ori,
                                only_upper_hemisphere=cfg.only_upper_hemisphere, num_layers=cfg.num_layers,
                                output_dim=cfg.output_dim, output_dim_vec=cfg.output_dim_vec, seed=seed)
    if cfg.aggr == "AttentionalAggregation" and cfg.model == "hepi":  # conv.py:21-26: gate_nn = Sequential(Linear(C, C), ReLU())
        gen_g = torch.Generator().manual_seed(seed + 555)
        for k, r in enumerate(rounds):
            for et in r:
                pre = f"processor.{k}.convs.{eq.conv_key(et)}.aggr_module.gate_nn.0"
                g[pre + ".weight"], g[pre + ".bias"] = eq.init_linear(64, 64, True, gen_g)
    actor = {"gnn." + k: v for k, v in g.items()}
    a_per = cfg.output_dim_vec * 3
    gen = torch.Generator().manual_seed(seed + 100)
    for name in ["_mean", "_pre_std"]:  # gnn_gaussian_policy_diag.py:21-24: orthogonal gain 0.01, zero bias
        w = torch.empty(a_per, 64)
        torch.nn.init.orthogonal_(w, 0.01, generator=gen)
        actor[f"{name}.weight"], actor[f"{name}.bias"] = w, torch.zeros(a_per)
    critic = gr.init_critic_params(len(spec.node_types) + 3 * spec.n_vec, seed=seed + 1)
    return actor, critic
