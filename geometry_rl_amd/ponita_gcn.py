"""EMPN actor GNN (= PonitaGCN around the homogeneous PONITA core) on the HIP kernels -- drop-in for
``geometry_rl/modules/pyg_models/ponita_gcn.py`` + ``ponita/ponita.py`` (Ponita, SeparableFiberBundleConvNext).

The reference flattens the hetero batch into ONE homogeneous graph (ponita_gcn.py:73-83,102-126) and runs every layer over
every edge.  Summation is linear, so the same result is obtained without materialising the flattened graph: per layer and
destination node type, the spatial conv x1 is the sum of the per-edge-type fused edge kernels (all sharing that layer's
weights), followed by the fiber conv and the ConvNeXt block of that layer.  Padded points have no edges and never reach the
read-out, so they are dropped exactly as in the HEPi graph (geometry_rl_amd/graph.py)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph import GraphBatch
from .hepi import basis_sequential, global_std, make_grid


class SeparableFiberBundleConv(nn.Module):
    """Parameter holder: ponita.py:100-147 (depthwise, no attention)."""

    def __init__(self, channels, kernel_dim):
        super().__init__()
        self.kernel = nn.Linear(kernel_dim, channels, bias=False)
        self.fiber_kernel = nn.Linear(kernel_dim, channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer("callibrated", torch.tensor(False))


class SeparableFiberBundleConvNext(nn.Module):
    """Parameter holder: ponita.py:195-217 with layer_scale=None."""

    def __init__(self, channels, kernel_dim, widening_factor=4):
        super().__init__()
        self.conv = SeparableFiberBundleConv(channels, kernel_dim)
        self.linear_1 = nn.Linear(channels, widening_factor * channels)
        self.linear_2 = nn.Linear(widening_factor * channels, channels)
        self.register_buffer("layer_scale", None)
        self.norm = nn.LayerNorm(channels)


class Ponita(nn.Module):
    """Parameter holder with the reference layout (ponita.py:247-325)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, output_dim_vec=0, dim=3, num_ori=16, degree=2,
                 widening_factor=4, only_upper_hemisphere=False, **ignored):
        super().__init__()
        self.dim, self.num_ori = dim, num_ori
        self.register_buffer("ori_grid", make_grid(dim, num_ori, only_upper_hemisphere))
        self.basis_fn = basis_sequential(14, hidden_dim, hidden_dim, degree)
        self.fiber_basis_fn = basis_sequential(3, hidden_dim, hidden_dim, degree)
        self.x_embedder = nn.Linear(input_dim, hidden_dim, False)
        self.interaction_layers = nn.ModuleList(
            [SeparableFiberBundleConvNext(hidden_dim, hidden_dim, widening_factor) for _ in range(num_layers)])
        self.read_out_layers = nn.ModuleList(
            [nn.Linear(hidden_dim, output_dim + output_dim_vec) if i == num_layers - 1 else None for i in range(num_layers)])


class PonitaGCN(nn.Module):
    def __init__(self, input_dim_node, output_dim, output_dim_vec, num_layers=2, hidden_dim=64, dropout=0.0, num_ori=16, degree=2,
                 widening_factor=4, attention=False, ponita_dim=3, only_upper_hemisphere=False, device="cuda", precision="fp32",
                 **ignored):
        super().__init__()
        self.precision, self._prec = precision, ("_bf16" if precision == "bf16" else "")
        if hidden_dim != 64 or num_ori != 16 or degree != 2 or widening_factor != 4 or attention:
            raise NotImplementedError("HIP kernels are specialised for configs/algorithm/pyg_agent/model/ponita_gcn.yaml")
        self.input_dim, self.dim = input_dim_node, ponita_dim
        self.output_dim, self.output_dim_vec = output_dim, output_dim_vec
        self.ponita = Ponita(input_dim_node, hidden_dim, output_dim, num_layers, output_dim_vec, ponita_dim, num_ori, degree,
                             widening_factor, only_upper_hemisphere)
        self.linear = nn.Linear(hidden_dim, output_dim + output_dim_vec)
        self.to(device)

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def decoder(self):  # the policy head reads the read-out layer through this name
        return self.linear

    @property
    def grid3(self):
        g = self.ponita.ori_grid
        return F.pad(g, (0, 3 - g.shape[1])).contiguous()

    @property
    def calibrated(self) -> bool:
        return all(bool(l.conv.callibrated) for l in self.ponita.interaction_layers)

    def _fiber_kernels(self):
        """fk = Phi Wf^T of every interaction layer from one fused launch (ops.FiberKernels); ponita.py:246-268."""
        poly = getattr(self, "_fiber_poly_cache", None)
        g = self.ponita.ori_grid
        if poly is None or poly.device != g.device:
            inv = (g[None, :, :] * g[:, None, :]).sum(-1, keepdim=True)
            poly = self._fiber_poly_cache = self.ponita.fiber_basis_fn[0](inv).detach().contiguous()
        return ops.fiber_kernels(poly, self.ponita.fiber_basis_fn, [l.conv for l in self.ponita.interaction_layers])

    def _layer(self, layer, x, graph: GraphBatch, grid3, fks, collect=None):
        b = self.ponita.basis_fn
        x1 = {}
        for et, es in graph.edges.items():  # spatial conv summed over all (merged) edge types: ponita.py:153,161
            s, _, d = et
            part = ops.EdgeConv.apply(x[s], graph.pos[s], graph.pos[d], grid3, b[1].weight, b[1].bias, b[3].weight, b[3].bias,
                                      layer.conv.kernel.weight, es, self.dim, None, self._prec)
            x1[d] = part if d not in x1 else x1[d] + part
        fk = fks[id(layer.conv)]
        out = {}
        for t, xt in x.items():
            x1t = x1.get(t)
            if x1t is None:
                x1t = torch.zeros_like(xt)
            x2 = ops.FiberConv.apply(x1t, fk, layer.conv.bias, self._prec)
            out[t] = ops.NodeMLP.apply(x2, xt, layer.norm.weight, layer.norm.bias, layer.linear_1.weight, layer.linear_1.bias,
                                       layer.linear_2.weight, layer.linear_2.bias, None, None, self._prec)
            if collect is not None:
                collect[t] = (x1t, fk)
        return out

    def latent_step(self, graph: GraphBatch, u_dict):
        scalar_dict, vector_dict = u_dict
        grid3 = self.grid3
        x = {t: ops.LiftEncode.apply(scalar_dict[t], vector_dict[t], grid3, self.ponita.x_embedder.weight, self._prec)
             for t in graph.node_types}
        fks = self._fiber_kernels()
        for layer in self.ponita.interaction_layers:
            x = self._layer(layer, x, graph, grid3, fks)
        lat = x[graph.output_mask_key]
        return lat.float() if lat.dtype != torch.float32 else lat

    def one_step(self, graph: GraphBatch, u_dict, **ignored):
        lat = self.latent_step(graph, u_dict)
        zw, zb = lat.new_zeros(3 * self.output_dim_vec, 64), lat.new_zeros(3 * self.output_dim_vec)
        mean, _, hidden = ops.Readout.apply(lat, self.grid3, self.linear.weight, self.linear.bias, zw, zb, 0.0, 0.0,
                                            self.output_dim, self.output_dim_vec)
        return mean.reshape(-1, 3), hidden

    @torch.no_grad()
    def calibrate(self, graph_full: GraphBatch, u_dict, group=None) -> None:
        """ponita.py:178-180,187-192: statistics over ALL nodes of the homogeneous graph (all node types, padding included)."""
        scalar_dict, vector_dict = u_dict
        grid3 = self.grid3
        x = {t: ops.LiftEncode.apply(scalar_dict[t], vector_dict[t], grid3, self.ponita.x_embedder.weight, self._prec)
             for t in graph_full.node_types}
        fks = self._fiber_kernels()
        cat = lambda d: torch.cat([d[t].reshape(-1) for t in graph_full.node_types])
        for layer in self.ponita.interaction_layers:
            col = {}
            out = self._layer(layer, x, graph_full, grid3, fks, collect=col)
            if not bool(layer.conv.callibrated):
                x1 = {t: col[t][0] for t in col}
                x2 = {t: ops.FiberConv.apply(col[t][0], col[t][1], torch.zeros_like(layer.conv.bias), self._prec) for t in col}
                s_in, s_1, s_2 = global_std(cat(x), group), global_std(cat(x1), group), global_std(cat(x2), group)
                layer.conv.kernel.weight.mul_(s_in / s_1)
                layer.conv.fiber_kernel.weight.mul_(s_1 / s_2)
                layer.conv.callibrated.fill_(True)
            x = out
