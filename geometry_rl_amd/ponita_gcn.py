"""EMPN actor GNN (= PonitaGCN around the homogeneous PONITA core) on the HIP kernels -- drop-in for
``geometry_rl/modules/pyg_models/ponita_gcn.py`` + ``ponita/ponita.py`` (Ponita, SeparableFiberBundleConvNext).

The reference flattens the hetero batch into ONE homogeneous graph (ponita_gcn.py:73-83,102-126) and runs every layer over
every edge (ponita.py:153-161).  So does this module (round 3): the node types share one latent array (read-out type last: the
per-type feature tensors of ``build_data`` are views of one buffer, nothing is concatenated), the edge types are merged once per
cached topology into ONE destination- / source-sorted edge set over that array, and a layer is one edge convolution, one fiber
convolution and one ConvNeXt block -- three launches forward, three backward (round 2: one edge launch per edge type plus torch
adds of the partial x1).  Padded points have no edges and never reach the read-out, so they are dropped exactly as in the HEPi
graph (geometry_rl_amd/graph.py).  For the same reason the LAST layer only computes what the read-out can see: its edge set holds
the edges into the read-out nodes, its fiber convolution and ConvNeXt block run on those nodes alone (the reference computes, and
then never reads, the last layer's output at every other node; its gradient is identically zero) -- ``prune_last_layer=False``
restores the full last layer."""
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
                 prune_last_layer=True, **ignored):
        super().__init__()
        self.prune_last_layer = prune_last_layer
        self._merged_cache = {}
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

    def _merged(self, graph: GraphBatch):
        """The homogeneous view of a cached topology: ONE edge set over the concatenated node array (ponita_gcn.py:73-83), and the
        sub edge set of the edges INTO the read-out nodes for the pruned last layer.  Built once per topology (host syncs allowed)."""
        hit = self._merged_cache.get(id(graph.edges))
        if hit is not None and hit[0] is graph.edges:
            return hit[1]
        order, off, n = graph.all_order, {}, 0
        for t in order:
            off[t] = n
            n += graph.num_nodes[t]
        srcs, dsts = [], []
        for (s_, _, d_), es in graph.edges.items():
            srcs.append(es.src_d.long() + off[s_])
            dsts.append(es.dst_d.long() + off[d_])
        src, dst = torch.cat(srcs), torch.cat(dsts)
        ro = graph.output_mask_key
        lo, n_ro = off[ro], graph.num_nodes[ro]
        assert order[-1] == ro and lo + n_ro == n
        into_ro = dst >= lo
        merged = dict(n=n, lo=lo, n_ro=n_ro, es_all=ops.build_edge_set(torch.stack([src, dst]), n, n),
                      es_last=ops.build_edge_set(torch.stack([src[into_ro], dst[into_ro] - lo]), n, n_ro))
        if len(self._merged_cache) >= 8:   # a handful of batch sizes / the padded calibration graph; never grows without bound
            self._merged_cache.pop(next(iter(self._merged_cache)))
        self._merged_cache[id(graph.edges)] = (graph.edges, merged)
        return merged

    def _weight_images(self, plan):
        """Pre-split weight images of the interaction layers of this pass from ONE launch (ops.weight_images); ``plan`` = [(layer, edge set)]."""
        b = self.ponita.basis_fn
        imgs = ops.weight_images([(l.conv.kernel.weight, es.n_dst, (l.norm.weight, l.norm.bias, l.linear_1.weight, l.linear_1.bias,
                                                                    l.linear_2.weight, l.linear_2.bias)) for l, es in plan],
                                 self.grid3, (b[1].weight, b[1].bias, b[3].weight, b[3].bias), self._prec,
                                 with_backward=torch.is_grad_enabled())
        return {id(l): im for (l, _), im in zip(plan, imgs)}

    def _layer_merged(self, layer, x, pos, es, grid3, fk, lo=0, collect=None, wimg=None):
        """One interaction layer on the homogeneous graph: x [N,16,64] -> [N_dst,16,64]; destinations = nodes lo .. (all, or the read-out tail)."""
        b = self.ponita.basis_fn
        x_dst, pos_dst = (x, pos) if lo == 0 else (x[lo:], pos[lo:])
        # x feeds the convolution AND the residual of its own node block: the two gradients are summed inside the fused edge backward
        res = {} if (lo == 0 and torch.is_grad_enabled() and x.requires_grad) else None
        x1 = ops.EdgeConv.apply(x, pos, pos_dst, grid3, b[1].weight, b[1].bias, b[3].weight, b[3].bias, layer.conv.kernel.weight, es,
                                self.dim, res, self._prec, wimg)
        x2 = ops.FiberConv.apply(x1, fk, layer.conv.bias, self._prec)
        if collect is not None:
            collect.update(x1=x1, fk=fk)
        return ops.NodeMLP.apply(x2, x_dst, layer.norm.weight, layer.norm.bias, layer.linear_1.weight, layer.linear_1.bias,
                                 layer.linear_2.weight, layer.linear_2.bias, None, res, self._prec, wimg)

    supports_head = True   # latent_step issues the merged head launch (ops.HeadLaunch) before its first consumer

    def latent_step(self, graph: GraphBatch, u_dict):
        grid3 = self.grid3
        mg = self._merged(graph)
        # parameter-only work first: with a head collector installed (ops.HEAD, policy.forward_diag) the fiber kernels and the weight images
        # share ONE launch with the node features build_data has handed over; the lift (which reads the features) comes behind it
        fks = self._fiber_kernels()
        layers = list(self.ponita.interaction_layers)
        last = lambda i: i == len(layers) - 1 and self.prune_last_layer
        wimgs = self._weight_images([(l, mg["es_last"] if last(i) else mg["es_all"]) for i, l in enumerate(layers)])
        if ops.HEAD is not None:
            ops.HEAD.launch(self._prec)
        x = ops.LiftEncode.apply(graph.scalar_all, graph.vec_all, grid3, self.ponita.x_embedder.weight, self._prec)
        for i, layer in enumerate(layers):
            if last(i):
                x = self._layer_merged(layer, x, graph.pos_all, mg["es_last"], grid3, fks[id(layer.conv)], lo=mg["lo"], wimg=wimgs[id(layer)])
            else:
                x = self._layer_merged(layer, x, graph.pos_all, mg["es_all"], grid3, fks[id(layer.conv)], wimg=wimgs[id(layer)])
        lat = x if x.shape[0] == mg["n_ro"] else x[mg["lo"]:]
        return lat.float() if lat.dtype != torch.float32 else lat

    def one_step(self, graph: GraphBatch, u_dict, **ignored):
        lat = self.latent_step(graph, u_dict)
        zw, zb = lat.new_zeros(3 * self.output_dim_vec, 64), lat.new_zeros(3 * self.output_dim_vec)
        mean, _, hidden = ops.Readout.apply(lat, self.grid3, self.linear.weight, self.linear.bias, zw, zb, 0.0, 0.0,
                                            self.output_dim, self.output_dim_vec)
        return mean.reshape(-1, 3), hidden

    @torch.no_grad()
    def calibrate(self, graph_full: GraphBatch, u_dict, group=None) -> None:
        """ponita.py:178-180,187-192: statistics over ALL nodes of the homogeneous graph (all node types, padding included; every layer
        in full -- no pruning here)."""
        grid3 = self.grid3
        mg = self._merged(graph_full)
        x = ops.LiftEncode.apply(graph_full.scalar_all, graph_full.vec_all, grid3, self.ponita.x_embedder.weight, self._prec)
        fks = self._fiber_kernels()
        for layer in self.ponita.interaction_layers:
            col = {}
            out = self._layer_merged(layer, x, graph_full.pos_all, mg["es_all"], grid3, fks[id(layer.conv)], collect=col)
            if not bool(layer.conv.callibrated):
                x1 = col["x1"]
                x2 = ops.FiberConv.apply(x1, col["fk"], torch.zeros_like(layer.conv.bias), self._prec)
                s_in, s_1, s_2 = global_std(x, group), global_std(x1, group), global_std(x2, group)
                layer.conv.kernel.weight.mul_(s_in / s_1)
                layer.conv.fiber_kernel.weight.mul_(s_1 / s_2)
                layer.conv.callibrated.fill_(True)
            x = out
