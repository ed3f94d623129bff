"""Policy / value heads around the HIP GNNs -- drop-ins for
``geometry_rl/algorithms/trust_region_projections/models/policy/gnn_gaussian_policy_diag.py`` (GNNGaussianPolicyDiag),
``.../models/value/gnn_vf_net.py`` (GNNVFNet), ``.../models/value/critic.py`` (BaseCritic) and
``geometry_rl/modules/pyg_models/deepsets.py`` (DeepSets).  Same constructor kwargs, forward signatures and state_dict names."""
import math
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .graph import HyperData


def inverse_softplus(x: torch.Tensor) -> torch.Tensor:
    """utils/torch_utils.py:361-370."""
    return (x.exp() - 1.0).log()


def _orthogonal_linear(in_f, out_f, gain):
    lin = nn.Linear(in_f, out_f)
    nn.init.orthogonal_(lin.weight, gain=gain)  # network_utils.py:66-70 ("orthogonal": weights orthogonal, biases zero)
    nn.init.zeros_(lin.bias)
    return lin


class GNNGaussianPolicyDiag(nn.Module):
    """forward(*obs, train=True) -> (loc [B,A], covariance_matrix [B,A,A] = diag(sigma)**2)   (gnn_gaussian_policy_diag.py:26-87).

    ``forward_diag`` is the allocation-free variant used by TRPLLoss: (loc [B,A], sigma [B,A])."""

    def __init__(self, gnn, hyper_data: HyperData, action_dim, num_actuators, init="orthogonal", hidden_sizes=(64, 64),
                 activation="tanh", layer_norm=False, contextual_std=True, trainable_std=True, init_std=1.0, use_tanh_mean=False,
                 share_weights=False, vf_model=None, minimal_std=1e-5, scale=1e-4, gain=0.01, share_action_dim=True, post_fc=False,
                 **kwargs):
        super().__init__()
        if not share_action_dim or isinstance(action_dim, list) or use_tanh_mean:
            raise NotImplementedError("policy head: share_action_dim=True, no tanh mean "
                                      "(configs/rigid_insertion_multi_hepi_trpl_cfg.yaml:85-99; post_fc and contextual_std either way)")
        if init != "orthogonal":
            raise NotImplementedError("only the 'orthogonal' initialisation of configs/algorithm/policy/default.yaml is mirrored")
        self.action_dim, self.num_actuators = action_dim, num_actuators
        self.contextual_std, self.post_fc = contextual_std, post_fc
        self.minimal_std = torch.tensor(minimal_std)
        self.init_std = torch.tensor(init_std)
        a_shared = action_dim // num_actuators
        self._pre_activation_shift = inverse_softplus(self.init_std - self.minimal_std)  # abstract_gaussian_policy.py:124-134
        self._mean = _orthogonal_linear(hidden_sizes[-1], a_shared, gain)  # unused when post_fc=False, kept for state_dict parity
        if contextual_std:
            self._pre_std = _orthogonal_linear(hidden_sizes[-1], a_shared, gain)
        else:   # a state-independent std: one trainable vector (gnn_gaussian_policy_diag.py:17-19, abstract_gnn_gaussian_policy.py:82-85)
            self._pre_std = nn.Parameter(torch.normal(0.0, 0.01, (a_shared,)))
            self.register_buffer("_zero_std_weight", torch.zeros(a_shared, hidden_sizes[-1]), persistent=False)
        if not trainable_std:   # abstract_gnn_gaussian_policy.py:84-85
            self._pre_std.requires_grad_(False)
        self.hyper_data = hyper_data
        self.gnn = gnn
        self._calib_checked = False  # host-side latch: the per-conv `callibrated` buffers are inspected once, not every step
        self.group = None            # torch.distributed group (data parallel): calibration statistics are summed over it
        self.to(next(gnn.parameters()).device)

    @property
    def is_diag(self):
        return True

    def _maybe_calibrate(self, args):
        """First training call: data-dependent re-initialisation of every conv that sees edges (conv.py:104-105).  Convs whose
        edge set is empty are skipped by the reference too (hetero_fiber_conv.py:48-49) and stay un-calibrated."""
        gnn = self.gnn
        self._calib_checked = True
        if hasattr(gnn, "calibrated") and not gnn.calibrated:
            hd = self.hyper_data
            full = HyperData(hd.spec, full_graph_obs=hd.full_graph_obs, dist_as_pos=hd.dist_as_pos,
                             output_mask_key=hd._output_mask_key, concat_input_vector=False, drop_padding=False)
            graph, u = full.build_data(*args, train=True)
            gnn.calibrate(graph, u, group=self.group)

    def load_state_dict(self, *a, **k):
        self._calib_checked = False
        return super().load_state_dict(*a, **k)

    def forward_diag(self, *args, train=True) -> Tuple[torch.Tensor, torch.Tensor]:
        self.train(train)
        B = args[0].shape[0]
        if self.post_fc:
            # gnn_gaussian_policy_diag.py:65-83 with post_fc=True (the default of abstract_gnn_gaussian_policy.py:37; config 1's
            # transformer actor): the GNN hands over ``hidden`` only and BOTH heads are Linear layers on it.  The GNN of this branch
            # is a stock torch module (geometry_rl_amd.transformer), so the two small heads stay torch ops under the same autograd.
            graph, u = self.hyper_data.build_data(*args, train=train)
            hidden = self.gnn.one_step(graph, u)
            mean = self._mean(hidden)
            shift = self._pre_activation_shift.to(hidden.device)
            pre = self._pre_std(hidden) if self.contextual_std else self._pre_std
            sigma = torch.nn.functional.softplus(pre + shift) + self.minimal_std.to(hidden.device)
            if not self.contextual_std:   # gnn_gaussian_policy_diag.py:73-74
                sigma = sigma.tile((hidden.shape[0], 1))
            return mean.reshape(B, -1), sigma.reshape(B, -1)
        if train and not self._calib_checked:
            self._maybe_calibrate(args)
        gnn = self.gnn
        # node features + fiber kernels + weight images of this pass as ONE launch (ops.HeadLaunch) where the GNN issues it
        ops.HEAD = ops.HeadLaunch() if (ops.FUSE_HEAD and getattr(gnn, "supports_head", False)) else None
        try:
            graph, u = self.hyper_data.build_data(*args, train=train)
            lat = gnn.latent_step(graph, u)
            if ops.HEAD is not None and (ops.HEAD.feat is not None or ops.HEAD.fiber is not None or ops.HEAD.wimg is not None):
                raise RuntimeError("the merged head launch was not issued by latent_step")
        finally:
            ops.HEAD = None
        dec = gnn.decoder
        # the std head of the fused read-out is softplus(Ws hidden + bs + shift) + minimal_std: a state-independent std is the same head with
        # Ws = 0 and bs = the trainable vector (gnn_gaussian_policy_diag.py:70-74: ``std = self._pre_std`` tiled over the nodes)
        ws, bs = (self._pre_std.weight, self._pre_std.bias) if self.contextual_std else (self._zero_std_weight, self._pre_std)
        mean, sigma, _ = ops.Readout.apply(lat, gnn.grid3, dec.weight, dec.bias, ws, bs,
                                           float(self._pre_activation_shift), float(self.minimal_std), gnn.output_dim,
                                           gnn.output_dim_vec)
        return mean.reshape(B, -1), sigma.reshape(B, -1)

    def forward(self, *args, train=True):
        loc, sigma = self.forward_diag(*args, train=train)
        return loc, sigma.diag_embed() ** 2

    # ---- diag-Gaussian helpers with the reference's "std matrix" API (gnn_gaussian_policy_diag.py:89-148)
    def sample(self, p, n=1):
        return self.rsample(p, n).detach()

    def rsample(self, p, n=1):
        means, std = p
        std = std.diagonal(dim1=-2, dim2=-1)
        eps = torch.randn((n,) + means.shape, dtype=std.dtype, device=std.device)
        return (means + eps * std).squeeze(0)

    def log_determinant(self, std):
        return 2 * std.diagonal(dim1=-2, dim2=-1).log().sum(-1)

    def maha(self, mean, mean_other, std):
        return ((mean - mean_other) / std.diagonal(dim1=-2, dim2=-1)).pow(2).sum(-1)

    def log_probability(self, p, x, **kwargs):
        mean, std = p
        k = x.shape[-1]
        return -0.5 * (self.maha(x, mean, std) + np.log(2.0 * np.pi) * k + self.log_determinant(std))

    def entropy(self, p):
        _, std = p
        return 0.5 * (std.shape[-1] * np.log(2 * np.e * np.pi) + self.log_determinant(std))

    def covariance(self, std):
        return std.pow(2)

    def precision(self, std):
        return (1 / self.covariance(std).diagonal(dim1=-2, dim2=-1)).diag_embed()

    def set_std(self, std: torch.Tensor) -> None:
        """gnn_gaussian_policy_diag.py:137-142: overwrite the state-independent std with the diagonal of ``std`` (a std MATRIX).  Written
        IN PLACE (the reference rebinds ``.data``): under PolicyUpdater the parameter is a view of the flat buffer and must stay one."""
        assert not self.contextual_std
        shifted_min = self.minimal_std + torch.finfo(std.dtype).eps   # avoid 0 on the diagonal: softplus^-1 fails there
        std_min = std.diagonal().clamp(min=shifted_min.to(std.device)) - self.minimal_std.to(std.device)
        new = inverse_softplus(std_min) - self._pre_activation_shift.to(std.device)
        if new.numel() != self._pre_std.numel():
            raise ValueError(f"set_std: {new.numel()} diagonal entries for a shared std of {self._pre_std.numel()} "
                             "(share_action_dim=True: one std per action dimension of ONE actuator)")
        self._pre_std.data.copy_(new.reshape(self._pre_std.shape).to(self._pre_std.device))


def get_policy_network(policy_type, proj_type, squash=False, device="cpu", dtype=torch.float32, **kwargs):
    """policy_factory.py:6-33 -- the call builders/utils_algo_graph.py:125-137 makes: ``policy_type == "gnn_diag"`` ->
    GNNGaussianPolicyDiag(**kwargs) on ``device``.  (``proj_type`` / ``squash`` are accepted and unused there as well.)"""
    if policy_type == "gnn_diag":
        policy = GNNGaussianPolicyDiag(**kwargs)
    else:
        raise ValueError(f"Invalid policy type {policy_type}. Select one of 'full', 'diag'.")
    if dtype not in (torch.float32, None):
        raise NotImplementedError("the HIP kernels store parameters in float32")
    return policy.to(device)


class _PygLinearNames(nn.Module):
    """PyG ``MLP([a, b, c], norm='layer_norm')`` parameter layout [upstream PyG 2.5.2]: lins.0, norms.0, lins.1."""

    def __init__(self, dims):
        super().__init__()
        self.lins = nn.ModuleList([nn.Linear(dims[0], dims[1]), nn.Linear(dims[1], dims[2])])
        self.norms = nn.ModuleList([nn.LayerNorm(dims[1])])  # used as whole-tensor ("graph" mode) LayerNorm by the kernels


class DeepSets(nn.Module):
    """deepsets.py:11-53 (norm = ['layer_norm', 'layer_norm'], configs/algorithm/pyg_agent/model/deepsets.yaml)."""

    def __init__(self, input_dim_node, output_dim=64, hidden_dim=64, norm=("layer_norm", "layer_norm"), device="cuda", **ignored):
        super().__init__()
        if hidden_dim != 64 or output_dim != 64 or list(norm) != ["layer_norm", "layer_norm"]:
            raise NotImplementedError("HIP DeepSets is specialised for hidden=output=64 with graph-mode LayerNorm")
        self.input_dim = input_dim_node
        self.mlp_inner = _PygLinearNames([input_dim_node, hidden_dim, hidden_dim])
        self.mlp_outer = _PygLinearNames([hidden_dim, hidden_dim, output_dim])
        self.to(device)

    @property
    def device(self):
        return next(self.parameters()).device


class GNNVFNet(nn.Module):
    """gnn_vf_net.py:8-102: critic GNN + Linear(64,1).  2-D inputs -> [B,1]; 3-D [N,T,.] inputs loop over T like the reference."""

    def __init__(self, gnn: DeepSets, hyper_data: HyperData, init="orthogonal", hidden_sizes=(64, 64), **kwargs):
        super().__init__()
        self.hyper_data, self.gnn = hyper_data, gnn
        self.final = nn.Linear(hidden_sizes[-1], 1)
        nn.init.orthogonal_(self.final.weight, 0.01)  # builders/utils_algo_graph.py:195-198
        nn.init.zeros_(self.final.bias)
        self.group = None  # torch.distributed group for the whole-batch LayerNorm statistics (data parallel)
        self.to(gnn.device)

    def _values(self, args, train):
        _, x = self.hyper_data.build_data(*args, train=train)
        g = self.gnn
        a, b = g.mlp_inner, g.mlp_outer
        return ops.DeepSetsValue.apply(x, a.lins[0].weight, a.lins[0].bias, a.norms[0].weight, a.norms[0].bias, a.lins[1].weight,
                                       a.lins[1].bias, b.lins[0].weight, b.lins[0].bias, b.norms[0].weight, b.norms[0].bias,
                                       b.lins[1].weight, b.lins[1].bias, self.final.weight, self.final.bias, self.group)

    # frames per grouped launch set: bounds the [frames, n, 64] fp32 intermediate of the first DeepSets stage to ~6 GB of the 288 GB
    GROUPED_BYTES = 6 << 30

    def _values_time_batched(self, args):
        """[N, T, .] inputs without autograd (the once-per-rollout critic pass, examples/torchrl/train.py:249-251): the T time steps as
        GROUPS of one launch set -- features of all frames from one ``grl_build_features`` launch, the three DeepSets stages once, with
        LayerNorm statistic slots per time step (the reference loops over T, gnn_vf_net.py:72-80: statistics per step).  Bitwise the loop's
        values; ~10 launches per chunk of time steps instead of ~10 per step.  Data parallel (``self.group``): the [T, slots] statistic
        arrays of a chunk are all-reduced ONCE per LayerNorm stage -- two collectives per chunk instead of two per time step."""
        N, T = args[0].shape[:2]
        g, (a, b) = self.gnn, (self.gnn.mlp_inner, self.gnn.mlp_outer)
        params = (a.lins[0].weight, a.lins[0].bias, a.norms[0].weight, a.norms[0].bias, a.lins[1].weight, a.lins[1].bias, b.lins[0].weight,
                  b.lins[0].bias, b.norms[0].weight, b.norms[0].bias, b.lins[1].weight, b.lins[1].bias, self.final.weight, self.final.bias)
        hd = self.hyper_data
        # rows per frame from a ONE-frame build, BEFORE anything rollout-sized is materialised (the chunk length used to be planned
        # after the features and the topology of all T * N frames existed -- 6.5 GB transient for cloth)
        _, x1 = hd.build_data(*[x[:1, 0] for x in args], train=False)
        n_nodes = x1.shape[1]
        step = max(1, min(T, int(self.GROUPED_BYTES // (N * n_nodes * 256))))
        out = torch.empty(N, T, device=args[0].device, dtype=torch.float32)
        keep = getattr(self, "_rollout_topo_key", None)
        for t0 in range(0, T, step):
            t1 = min(T, t0 + step)
            tm = [x[:, t0:t1].transpose(0, 1).reshape((t1 - t0) * N, -1) for x in args]   # time-major frames of this chunk
            key = (t1 - t0) * N
            fresh = key not in hd._cache
            _, xf = hd.build_data(*tm, train=False)
            out[:, t0:t1] = ops.deepsets_values_groups(xf, params, t1 - t0, group=self.group).transpose(0, 1)
            # the dense critic never reads a topology's edges, but a rollout-sized entry still holds index arrays of every frame: at most ONE
            # of them stays cached (the full chunk, reused by the next rollout); tails and re-planned chunk lengths are dropped
            if fresh and key != N and key != 1:
                if key == step * N and step * N != keep:
                    if keep is not None and keep not in (N, 1):
                        hd._cache.pop(keep, None)
                    keep = self._rollout_topo_key = key
                elif key != keep:
                    hd._cache.pop(key, None)
        return out

    def forward(self, *args, train=True):
        self.train(train)
        if args[0].dim() == 3:
            if not torch.is_grad_enabled():
                return self._values_time_batched(args).unsqueeze(-1)
            T = args[0].shape[1]
            return torch.stack([self._values([a[:, i] for a in args], train) for i in range(T)], dim=1).unsqueeze(-1)
        return self._values(args, train).unsqueeze(-1)


class BaseCritic(nn.Module):
    """critic.py:4-32."""

    def __init__(self, vf):
        super().__init__()
        self._network1 = vf

    def forward(self, x, *args, **kwargs):
        return self._network1(x, *args, **kwargs)

    def q1(self, x, *args, **kwargs):
        return self._network1(x, *args, **kwargs)

    @property
    def is_vf(self):
        return True


def get_critic(critic_type: str, dim: int = 0, device="cpu", dtype=torch.float32, **kwargs):
    """critic_factory.py:7-33 -- the call builders/utils_algo_graph.py:180-187 makes: ``critic_type == "gnn"`` ->
    BaseCritic(GNNVFNet(**kwargs)).  The builder's orthogonal re-initialisation of every Linear (utils_algo_graph.py:195-198) works on the
    returned module unchanged (its Linear children are ordinary ``nn.Linear`` parameter holders)."""
    if critic_type == "gnn":
        if dtype not in (torch.float32, None):
            raise NotImplementedError("the HIP kernels store parameters in float32")
        return BaseCritic(GNNVFNet(**kwargs)).to(device)
    raise ValueError(f"Invalid value_loss type {critic_type}. Select one of 'base', 'double', 'duelling'.")
