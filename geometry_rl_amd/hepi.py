"""HEPi actor GNN on the HIP kernels -- drop-in for ``geometry_rl/modules/pyg_models/hepi.py`` (+ ponita/conv.py,
ponita/hetero_fiber_conv.py).  Same constructor kwargs, same ``one_step(graph, u_dict)`` contract, same ``state_dict``
names (PyG's tuple-key mangling ``<src___rel___dst>`` included) so reference checkpoints load.

The nn.Linear / LayerNorm children are parameter containers only: their weights are handed to the fused HIP kernels
(geometry_rl_amd.ops); torch never runs them."""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph import EdgeType, GraphBatch


def make_grid(dim: int, n: int, only_upper_hemisphere: bool = False) -> torch.Tensor:
    """S1 uniform / S2 Fibonacci lattice orientation grid (reference ponita.py:53-97)."""
    if dim == 2:
        ang = torch.linspace(0, 2 * math.pi - (2 * math.pi / n), n)
        return torch.stack((torch.cos(ang), torch.sin(ang)), dim=1)
    if dim != 3:
        raise ValueError("Only S1 and S2 are supported.")
    i = torch.arange(n)
    theta = (math.pi * i * (1 + math.sqrt(5))) % (2 * math.pi)
    phi = torch.acos(1 - (1.0 if only_upper_hemisphere else 2.0) * (i + 0.5) / (n - 1 + 1.0))
    return torch.stack((torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)), dim=-1)


class PolynomialFeatures(nn.Module):
    """Parameter-free placeholder so ``basis_fn.1`` / ``basis_fn.3`` keep the reference indices (ponita.py:233-244)."""

    def __init__(self, degree):
        super().__init__()
        self.degree = degree

    def forward(self, x):
        polys = [x]
        for _ in range(self.degree):
            polys.append(torch.einsum("...i,...j->...ij", polys[-1], x).flatten(-2, -1))
        return torch.cat(polys, -1)


def basis_sequential(in_dim: int, hidden: int, basis: int, degree: int) -> nn.Sequential:
    return nn.Sequential(PolynomialFeatures(degree), nn.Linear(in_dim, hidden), nn.GELU(), nn.Linear(hidden, basis), nn.GELU())


class FiberBundleConv(nn.Module):
    """Parameter holder with the reference's names/shapes (conv.py:10-69); separable + depthwise only."""

    def __init__(self, in_channels, out_channels, attr_dim, bias=True, aggr="add", separable=True, groups=1, widening_factor=4):
        super().__init__()
        if not (separable and groups == in_channels == out_channels == 64 and attr_dim == 64 and widening_factor == 4 and bias):
            raise NotImplementedError("the HIP path implements the separable depth-wise 64-channel configuration of every "
                                      "reference config (configs/algorithm/pyg_agent/model/hepi.yaml:19-48)")
        if aggr not in ("add", "AttentionalAggregation"):
            raise NotImplementedError("aggr: 'add' (hepi.yaml) or 'AttentionalAggregation' (hepi_attention.yaml)")
        self.attention = aggr == "AttentionalAggregation"
        if self.attention:   # conv.py:21-26: PyG registers the aggregation as ``aggr_module`` with gate_nn = Sequential(Linear, ReLU)
            self.aggr_module = nn.Module()
            self.aggr_module.gate_nn = nn.Sequential(nn.Linear(in_channels, in_channels), nn.ReLU())
        self.kernel = nn.Linear(attr_dim, in_channels, bias=False)
        self.fiber_kernel = nn.Linear(attr_dim, in_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.register_buffer("callibrated", torch.tensor(False))
        self.node_mlp = nn.Sequential(nn.LayerNorm(in_channels), nn.Linear(in_channels, out_channels * widening_factor), nn.GELU(),
                                      nn.Linear(out_channels * widening_factor, out_channels))


def global_std(t: torch.Tensor, group=None) -> torch.Tensor:
    """Unbiased std over ALL elements of ``t`` (conv.py:151-157 uses ``tensor.std()``); with a process group the three moments are
    summed over the ranks first, so every data-parallel replica computes the calibration factors of the WHOLE minibatch -- the same
    numbers a single device would (no rank-local re-initialisation, no divergence between replicas)."""
    if group is None:
        return t.float().std()   # (bf16 latents: statistics in fp32)
    import torch.distributed as dist
    td = t.double()
    m = torch.stack([td.sum(), (td * td).sum(), torch.tensor(float(t.numel()), dtype=torch.float64, device=t.device)])
    dist.all_reduce(m, group=group)
    n, mean = m[2], m[0] / m[2]
    return ((m[1] - n * mean * mean) / (n - 1)).clamp_min(0).sqrt().float()


def conv_key(edge_type: EdgeType) -> str:
    return "<" + "___".join(edge_type) + ">"


class HeteroFiberConv(nn.Module):
    """One message-passing round: a FiberBundleConv per edge type, outputs summed per destination type
    (hetero_fiber_conv.py:10-66)."""

    def __init__(self, convs: Dict[EdgeType, FiberBundleConv]):
        super().__init__()
        self.edge_types = [tuple(k) for k in convs]
        self.convs = nn.ModuleDict({conv_key(tuple(k)): v for k, v in convs.items()})

    def items(self):
        return [(et, self.convs[conv_key(et)]) for et in self.edge_types]


class HEPi(nn.Module):
    supports_head = True   # latent_step issues the merged head launch (ops.HeadLaunch) before its first consumer

    def __init__(self, input_dim_node, input_dim_edge, hidden_dim, latent_dim, output_dim, output_dim_vec, node_encoder_layers=2,
                 edge_encoder_layers=2, node_decoder_layers=2, node_type_mapping=None, edge_type_mapping=None,
                 edge_level_mapping=None, message_passing=None, num_messages=2, concat_global=False, shared_processor=False,
                 shared_node_encoder=True, shared_edge_encoder=True, device="cuda", num_ori=16, basis_dim=None, degree=2,
                 ponita_dim=3, only_upper_hemisphere=False, precision="fp32", **ignored):
        super().__init__()
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision: 'fp32' (split-bf16 products, fp32-accurate) or 'bf16' (one bf16 MFMA per product)")
        self.precision, self._prec = precision, ("_bf16" if precision == "bf16" else "")
        if hidden_dim != 64 or latent_dim != 64 or num_ori != 16 or degree != 2 or concat_global or shared_processor:
            raise NotImplementedError("HIP kernels are specialised for hidden=latent=64, 16 orientations, degree 2")
        self.input_dim_node, self.output_dim, self.output_dim_vec = input_dim_node, output_dim, output_dim_vec
        self.latent_dim, self.hidden_dim, self.num_messages = latent_dim, hidden_dim, num_messages
        self.device = device
        self.dim, self.num_ori = ponita_dim, num_ori
        self.register_buffer("ori_grid", make_grid(ponita_dim, num_ori, only_upper_hemisphere))
        self.basis_fn = basis_sequential(14, hidden_dim, hidden_dim, degree)
        self.fiber_basis_fn = basis_sequential(3, hidden_dim, hidden_dim, degree)
        self.node_encoder = nn.Linear(input_dim_node, latent_dim, False)
        self.processor = nn.ModuleList()
        for k in range(num_messages):  # hepi.py:93-104
            level = {}
            for l, edge_level in enumerate(edge_level_mapping):
                pl = message_passing[l][k]
                if pl is not None:
                    for et in edge_type_mapping:
                        if et[1] == edge_level:
                            level[tuple(et)] = pl
            self.processor.append(HeteroFiberConv(level))
        self.decoder = nn.Linear(latent_dim, output_dim + output_dim_vec)
        self.to(device)

    # ------------------------------------------------------------------ helpers
    @property
    def grid3(self) -> torch.Tensor:
        """Orientation grid padded to 3 columns (constant: built once, no per-call launches)."""
        g3 = getattr(self, "_grid3_cache", None)
        if g3 is None or g3.device != self.ori_grid.device:
            g = self.ori_grid
            g3 = self._grid3_cache = F.pad(g, (0, 3 - g.shape[1])).contiguous()
        return g3

    @property
    def calibrated(self) -> bool:
        return all(bool(c.callibrated) for r in self.processor for _, c in r.items())

    def fiber_poly(self) -> torch.Tensor:
        """Polynomial features of the (constant) grid invariants o_o . o_p  (hepi.py:119): [16,16,3], built once."""
        poly = getattr(self, "_fiber_poly_cache", None)
        if poly is None or poly.device != self.ori_grid.device:
            g = self.ori_grid
            inv = (g[None, :, :] * g[:, None, :]).sum(-1, keepdim=True)
            poly = self._fiber_poly_cache = self.fiber_basis_fn[0](inv).detach().contiguous()
        return poly

    def fiber_basis(self) -> torch.Tensor:
        """Phi[o,p,:] = fiber_basis_fn(o_o . o_p)  (hepi.py:119,157) in plain torch (inspection / tests)."""
        return self.fiber_basis_fn[1:](self.fiber_poly())

    def _fiber_kernels(self, graph: GraphBatch):
        """fk = Phi Wf^T of every convolution this pass will run, from one fused launch (ops.FiberKernels)."""
        convs = [conv for rnd in self.processor for et, conv in rnd.items() if et in graph.edges]
        return ops.fiber_kernels(self.fiber_poly(), self.fiber_basis_fn, convs)

    def _needed_types(self, graph: GraphBatch):
        need = {graph.output_mask_key} if graph.output_mask_key else set(graph.node_types)
        for r in self.processor:
            for (s, _, d), _c in r.items():
                if (s, _, d) in graph.edges:
                    need.update((s, d))
        return [t for t in graph.node_types if t in need]

    def _weight_images(self, graph: GraphBatch):
        """Pre-split weight images of every convolution block this pass will run, from ONE launch (ops.weight_images): the MFMA kernels
        copy them instead of re-splitting the weights in each of their workgroups' prologues; the backward pass reuses them."""
        items = [(et, conv) for rnd in self.processor for et, conv in rnd.items() if et in graph.edges and not getattr(conv, "attention", False)]
        b = self.basis_fn
        imgs = ops.weight_images([(c.kernel.weight, graph.edges[et].n_dst,
                                   (c.node_mlp[0].weight, c.node_mlp[0].bias, c.node_mlp[1].weight, c.node_mlp[1].bias, c.node_mlp[3].weight,
                                    c.node_mlp[3].bias)) for et, c in items],
                                 self.grid3, (b[1].weight, b[1].bias, b[3].weight, b[3].bias), self._prec,
                                 with_backward=torch.is_grad_enabled())
        return {id(c): im for (et, c), im in zip(items, imgs)}

    def _conv(self, conv, x_src, x_dst, graph, et, grid3, fks, prev, wimgs=None):
        es = graph.edges[et]
        s, _, d = et
        b = self.basis_fn
        wimg = wimgs.get(id(conv)) if wimgs else None
        # x feeds the convolution AND the residual of its own node block: the two gradients are summed inside the d x_src kernel
        res = {} if (x_src is x_dst and prev is None and torch.is_grad_enabled() and x_src.requires_grad) else None
        if getattr(conv, "attention", False):
            # messages per edge -> gate network (a plain library GEMM + ReLU, autograd) -> per-destination softmax-weighted sum
            msg = ops.EdgeMessages.apply(x_src, graph.pos[s], graph.pos[d], grid3, b[1].weight, b[1].bias, b[3].weight, b[3].bias,
                                         conv.kernel.weight, es, self.dim, res, self._prec)
            gate = conv.aggr_module.gate_nn(msg if msg.dtype == torch.float32 else msg.float())
            x1 = ops.SoftmaxAggregate.apply(gate, msg, es, self._prec)
        else:
            x1 = ops.EdgeConv.apply(x_src, graph.pos[s], graph.pos[d], grid3, b[1].weight, b[1].bias, b[3].weight, b[3].bias,
                                    conv.kernel.weight, es, self.dim, res, self._prec, wimg)
        fk = fks[id(conv)]
        x2 = ops.FiberConv.apply(x1, fk, conv.bias, self._prec)
        m = conv.node_mlp
        return ops.NodeMLP.apply(x2, x_dst, m[0].weight, m[0].bias, m[1].weight, m[1].bias, m[3].weight, m[3].bias, prev, res,
                                 self._prec, wimg), x1, fk

    # ------------------------------------------------------------------ forward
    def latent_step(self, graph: GraphBatch, u_dict) -> torch.Tensor:
        """hepi.py:125-173: lift/encode, message-passing rounds; returns the actuator latents [B*G, 16, 64]."""
        scalar_dict, vector_dict = u_dict
        grid3 = self.grid3
        types = self._needed_types(graph)
        # parameter-only work of the pass (fiber kernels, pre-split weight images) is issued FIRST: with a head collector installed
        # (ops.HEAD, policy.forward_diag) it shares ONE launch with the node features that build_data has handed over
        fks = self._fiber_kernels(graph)
        wimgs = self._weight_images(graph)
        if ops.HEAD is not None:
            ops.HEAD.launch(self._prec)
        if 1 < len(types) <= 4:   # every node type in ONE lift launch (each way)
            xs = ops.LiftEncodeMulti.apply(grid3, self.node_encoder.weight, self._prec, *[a for t in types for a in (scalar_dict[t], vector_dict[t])])
            x = dict(zip(types, xs))
        else:
            x = {t: ops.LiftEncode.apply(scalar_dict[t], vector_dict[t], grid3, self.node_encoder.weight, self._prec) for t in types}
        for rnd in self.processor:
            outs = {}
            for et, conv in rnd.items():
                if et not in graph.edges:  # empty edge set: skipped like hetero_fiber_conv.py:48-49
                    continue
                s, _, d = et
                outs[d], _, _ = self._conv(conv, x[s], x[d], graph, et, grid3, fks, outs.get(d), wimgs)
            x.update(outs)
        lat = x[graph.output_mask_key]
        return lat.float() if lat.dtype != torch.float32 else lat   # the read-out of the few actuator nodes runs in fp32

    def one_step(self, graph: GraphBatch, u_dict, u=None, u_properties=None):
        """Reference contract: -> (out [B*G*out_vec, 3], hidden [B*G, 64])."""
        lat = self.latent_step(graph, u_dict)
        zw = lat.new_zeros(3 * self.output_dim_vec, 64)
        zb = lat.new_zeros(3 * self.output_dim_vec)
        mean, _, hidden = ops.Readout.apply(lat, self.grid3, self.decoder.weight, self.decoder.bias, zw, zb, 0.0, 0.0,
                                            self.output_dim, self.output_dim_vec)
        return mean.reshape(-1, 3), hidden

    @torch.no_grad()
    def calibrate(self, graph_full: GraphBatch, u_dict, group=None) -> None:
        """First-training-call re-initialisation (conv.py:104-105,151-157) on the FULL (padded) graph: kernel.weight *=
        std(x_dst)/std(x_1), fiber_kernel.weight *= std(x_1)/std(x_2) with x_2 taken before the bias; the call that
        calibrates continues with the un-rescaled activations, so later rounds see the same inputs as in the reference.
        ``group``: data-parallel process group -- the statistics are then those of the whole (sharded) minibatch."""
        scalar_dict, vector_dict = u_dict
        grid3 = self.grid3
        x = {t: ops.LiftEncode.apply(scalar_dict[t], vector_dict[t], grid3, self.node_encoder.weight, self._prec)
             for t in graph_full.node_types}
        fks = self._fiber_kernels(graph_full)
        for rnd in self.processor:
            outs = {}
            for et, conv in rnd.items():
                if et not in graph_full.edges:
                    continue
                s, _, d = et
                out, x1, fk = self._conv(conv, x[s], x[d], graph_full, et, grid3, fks, outs.get(d))
                if not bool(conv.callibrated):
                    x2 = ops.FiberConv.apply(x1, fk, torch.zeros_like(conv.bias), self._prec)
                    s_in, s_1, s_2 = global_std(x[d], group), global_std(x1, group), global_std(x2, group)
                    conv.kernel.weight.mul_(s_in / s_1)
                    conv.fiber_kernel.weight.mul_(s_1 / s_2)
                    conv.callibrated.fill_(True)
                outs[d] = out
            x.update(outs)
