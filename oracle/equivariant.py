"""Oracle: HEPi / EMPN equivariant message-passing actor (plain torch, CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Every function cites the reference lines it restates.  Abbreviations:
  hepi.py      = geometry_rl/modules/pyg_models/hepi.py
  conv.py      = geometry_rl/modules/pyg_models/ponita/conv.py
  hetero.py    = geometry_rl/modules/pyg_models/ponita/hetero_fiber_conv.py
  ponita.py    = geometry_rl/modules/pyg_models/ponita/ponita.py
  ponita_gcn.py= geometry_rl/modules/pyg_models/ponita_gcn.py
  sphere.py    = geometry_rl/modules/pyg_models/ponita/utils/to_from_sphere.py

Parameters are passed as flat dicts keyed by the reference ``state_dict`` names
(relative to the gnn module), so golden fixtures and the product modules can
share weights by name.
"""
import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

EdgeType = Tuple[str, str, str]


# --------------------------------------------------------------------------- grids
def make_grid(dim: int, n: int, only_upper_hemisphere: bool = False, dtype=torch.float32) -> torch.Tensor:
    """ponita.py:53-97 GridGenerator (S1 uniform angles / S2 Fibonacci lattice)."""
    if dim == 2:
        # ponita.py:53-57
        angles = torch.linspace(0, 2 * math.pi - (2 * math.pi / n), n)
        return torch.stack((torch.cos(angles), torch.sin(angles)), dim=1).to(dtype)
    if dim != 3:
        raise ValueError("Only S1 and S2 are supported.")
    # ponita.py:65-97 (offset = 0.5)
    i = torch.arange(n)
    theta = (math.pi * i * (1 + math.sqrt(5))) % (2 * math.pi)
    scale = 1.0 if only_upper_hemisphere else 2.0
    phi = torch.acos(1 - scale * (i + 0.5) / (n - 1 + 2 * 0.5))
    return torch.stack(
        (torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)), dim=-1
    ).to(dtype)


# --------------------------------------------------------------------------- lift
def scalar_to_sphere(scalar: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """sphere.py:8-9."""
    return scalar.unsqueeze(-2).repeat_interleave(grid.shape[-2], dim=-2)


def vec_to_sphere(vec: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """sphere.py:4-5: [N,V,d] x [O,d] -> [N,O,V]."""
    return torch.einsum("bcd,nd->bnc", vec, grid)


def lift_features(scalars: torch.Tensor, vectors: torch.Tensor, grid: torch.Tensor, dim: int) -> torch.Tensor:
    """hepi.py:136-142: scalars [N,S], vectors [N,3V] -> [N,O,S+V]."""
    s = scalar_to_sphere(scalars, grid)
    v = vectors.view(s.shape[0], -1, 3)
    v = v[..., :2] if dim == 2 else v
    v = vec_to_sphere(v, grid)
    return torch.cat([s, v], dim=-1)


# --------------------------------------------------------------------------- invariants / bases
def polynomial_features(x: torch.Tensor, degree: int = 2) -> torch.Tensor:
    """ponita.py:233-244 PolynomialFeatures."""
    polys = [x]
    for _ in range(degree):
        polys.append(torch.einsum("...i,...j->...ij", polys[-1], x).flatten(-2, -1))
    return torch.cat(polys, -1)


def spatial_invariants(grid: torch.Tensor, pos_send: torch.Tensor, pos_receive: torch.Tensor) -> torch.Tensor:
    """hepi.py:109-123 (identical maths: ponita.py:327-345): [E,O,2]."""
    rel = (pos_send - pos_receive)[:, None, :]
    ga = grid[None, :, :]
    inv1 = (rel * ga).sum(dim=-1, keepdim=True)
    inv2 = (rel - inv1 * ga).norm(dim=-1, keepdim=True)
    return torch.cat([inv1, inv2], dim=-1)


def orientation_invariants(grid: torch.Tensor) -> torch.Tensor:
    """hepi.py:119 / ponita.py:339: [O,O,1]."""
    return (grid[None, :, :] * grid[:, None, :]).sum(dim=-1, keepdim=True)


def basis_mlp(x: torch.Tensor, P: Dict[str, torch.Tensor], prefix: str, degree: int = 2) -> torch.Tensor:
    """hepi.py:76-89 basis_fn / fiber_basis_fn: Poly -> Linear -> GELU -> Linear -> GELU (exact erf GELU)."""
    h = polynomial_features(x, degree)
    h = F.gelu(F.linear(h, P[f"{prefix}.1.weight"], P[f"{prefix}.1.bias"]))
    h = F.gelu(F.linear(h, P[f"{prefix}.3.weight"], P[f"{prefix}.3.bias"]))
    return h


# --------------------------------------------------------------------------- one conv layer
def scatter_sum(msg: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """torch_scatter.scatter(reduce="sum") as used at conv.py:141-147 (== ponita.py:11-18)."""
    out = torch.zeros((dim_size,) + tuple(msg.shape[1:]), dtype=msg.dtype)
    return out.index_add_(0, index, msg)


def attentional_aggregation(msg: torch.Tensor, index: torch.Tensor, dim_size: int, w_gate: torch.Tensor, b_gate: torch.Tensor):
    """PyG 2.5.2 AttentionalAggregation(gate_nn=Sequential(Linear, ReLU)) [upstream; PARITY UNPINNED beyond the call site], vmapped over
    the orientation axis at conv.py:58-61 (the gate network acts on the channel axis, so the vmap is a plain broadcast):
    gate = ReLU(Linear(msg)); alpha = utils.softmax(gate, index) = exp(gate - max_group) / (sum_group + 1e-16) with the max detached;
    out = scatter_sum(alpha * msg)."""
    gate = F.relu(F.linear(msg, w_gate, b_gate))
    mx = torch.full((dim_size,) + tuple(gate.shape[1:]), -float("inf"), dtype=gate.dtype)
    mx = mx.scatter_reduce(0, index.reshape(-1, *([1] * (gate.dim() - 1))).expand_as(gate), gate.detach(), "amax", include_self=True)
    ex = (gate - mx[index]).exp()
    den = scatter_sum(ex, index, dim_size) + 1e-16
    return scatter_sum(ex / den[index] * msg, index, dim_size)


def fiber_bundle_conv(
    x_src: torch.Tensor,
    x_dst: torch.Tensor,
    edge_index: torch.Tensor,
    kernel_basis: torch.Tensor,
    fiber_basis: torch.Tensor,
    P: Dict[str, torch.Tensor],
    prefix: str,
    return_intermediates: bool = False,
):
    """conv.py:71-113 FiberBundleConv.forward (separable, depthwise) for (x_src, x_dst).

    PyG propagate restated from the call site conv.py:80-86,115-117,128-149:
    x_j = x_src[edge_index[0]], message = kernel * x_j, aggregate = scatter-sum over
    edge_index[1] with dim_size = |dst|.
    """
    kernel = F.linear(kernel_basis, P[f"{prefix}.kernel.weight"])  # conv.py:79
    msg = kernel * x_src[edge_index[0]]  # conv.py:115-117
    if f"{prefix}.aggr_module.gate_nn.0.weight" in P:  # aggr="AttentionalAggregation" (conv.py:21-26,58-61,138-139)
        x_1 = attentional_aggregation(msg, edge_index[1].long(), x_dst.shape[0], P[f"{prefix}.aggr_module.gate_nn.0.weight"],
                                      P[f"{prefix}.aggr_module.gate_nn.0.bias"])
    else:
        x_1 = scatter_sum(msg, edge_index[1].long(), x_dst.shape[0])  # conv.py:141-147
    fiber_kernel = F.linear(fiber_basis, P[f"{prefix}.fiber_kernel.weight"])  # conv.py:88
    x_2 = torch.einsum("boc,opc->bpc", x_1, fiber_kernel) / fiber_kernel.shape[-2]  # conv.py:90
    x_2b = x_2 + P[f"{prefix}.bias"]  # conv.py:108-109
    # conv.py:64-69,112: LayerNorm -> Linear -> GELU -> Linear, residual on x_dst
    h = F.layer_norm(x_2b, (x_2b.shape[-1],), P[f"{prefix}.node_mlp.0.weight"], P[f"{prefix}.node_mlp.0.bias"], 1e-5)
    h = F.gelu(F.linear(h, P[f"{prefix}.node_mlp.1.weight"], P[f"{prefix}.node_mlp.1.bias"]))
    h = F.linear(h, P[f"{prefix}.node_mlp.3.weight"], P[f"{prefix}.node_mlp.3.bias"])
    out = x_dst + h
    if return_intermediates:
        return out, x_1, x_2
    return out


def calibrate_conv(x_dst, x_1, x_2, P: Dict[str, torch.Tensor], prefix: str) -> None:
    """conv.py:104-105,151-157: one-shot data-dependent rescale (unbiased std over all elements,
    x_2 taken BEFORE the bias)."""
    with torch.no_grad():
        std_in, std_1, std_2 = x_dst.std(), x_1.std(), x_2.std()
        P[f"{prefix}.kernel.weight"] = P[f"{prefix}.kernel.weight"] * std_in / std_1
        P[f"{prefix}.fiber_kernel.weight"] = P[f"{prefix}.fiber_kernel.weight"] * std_1 / std_2


# --------------------------------------------------------------------------- HEPi
def conv_key(edge_type: EdgeType) -> str:
    """PyG 2.5.2 ModuleDict key mangling for tuple keys [upstream]: '<a___b___c>'."""
    return "<" + "___".join(edge_type) + ">"


def hepi_schedule(edge_types: List[EdgeType], edge_levels: List[str], codes: List[List[int]]) -> List[List[EdgeType]]:
    """hepi.py:93-104 + builders/utils_algo_graph.py:34-47: round k holds every edge type whose
    level has code[level][k] == 1, in edge_level order then edge_type enum order."""
    num_messages = len(codes[0])
    rounds = []
    for k in range(num_messages):
        r = []
        for l, level in enumerate(edge_levels):
            if codes[l][k] == 1:
                r.extend([et for et in edge_types if et[1] == level])
        rounds.append(r)
    return rounds


def hepi_forward(
    P: Dict[str, torch.Tensor],
    graph: dict,
    scalar_dict: Dict[str, torch.Tensor],
    vector_dict: Dict[str, torch.Tensor],
    *,
    dim: int,
    output_dim: int,
    output_dim_vec: int,
    rounds: List[List[EdgeType]],
    calibrate: bool = False,
):
    """hepi.py:125-190 HEPi.one_step -> (out [B*G*out_vec, 3], hidden [B*G, C]).

    ``graph``: {"node_types": [...], "pos": {t: [N_t,3]}, "edge_index": {et: [2,E]},
    "output_mask_key": t}.  ``calibrate=True`` reproduces the first training call
    (conv.py:104-105) and updates P in place.
    """
    grid = P["ori_grid"]
    num_ori = grid.shape[0]
    latent = {}
    for t in graph["node_types"]:  # hepi.py:136-143
        x = lift_features(scalar_dict[t], vector_dict[t], grid, dim)
        latent[t] = F.linear(x, P["node_encoder.weight"])

    kernel_basis, fiber_basis = {}, {}
    for et, ei in graph["edge_index"].items():  # hepi.py:145-157
        src, _, dst = et
        ps = graph["pos"][src][ei[0]]
        pd = graph["pos"][dst][ei[1]]
        if dim == 2:
            ps, pd = ps[..., :2], pd[..., :2]
        kernel_basis[et] = basis_mlp(spatial_invariants(grid, ps, pd), P, "basis_fn")
        fiber_basis[et] = basis_mlp(orientation_invariants(grid), P, "fiber_basis_fn")

    for k, round_types in enumerate(rounds):  # hepi.py:164-171 -> hetero.py:31-66
        outs: Dict[str, List[torch.Tensor]] = {}
        for et in round_types:
            src, _, dst = et
            ei = graph["edge_index"].get(et)
            if ei is None or ei.numel() == 0:  # hetero.py:48-49
                continue
            prefix = f"processor.{k}.convs.{conv_key(et)}"
            if calibrate:
                # conv.py:104-112: the first training call rescales the weights AFTER x_1/x_2 were
                # computed, and still finishes that call with the un-rescaled x_2.
                o, x1, x2 = fiber_bundle_conv(
                    latent[src], latent[dst], ei, kernel_basis[et], fiber_basis[et], P, prefix, True
                )
                calibrate_conv(latent[dst], x1, x2, P, prefix)
            else:
                o = fiber_bundle_conv(latent[src], latent[dst], ei, kernel_basis[et], fiber_basis[et], P, prefix)
            outs.setdefault(dst, []).append(o)
        for t, vals in outs.items():  # hetero.py:63-64 group(..., "sum")
            latent[t] = torch.stack(vals, 0).sum(0) if len(vals) > 1 else vals[0]

    lat = latent[graph["output_mask_key"]]  # hepi.py:173
    return readout(lat, P["decoder.weight"], P["decoder.bias"], grid, dim, output_dim, output_dim_vec)


def readout(lat, w, b, grid, dim, output_dim, output_dim_vec):
    """hepi.py:180-190 (same lines in ponita_gcn.py:129-146)."""
    num_ori = grid.shape[0]
    y = F.linear(lat, w, b)
    out_scalar, out_vec = y.split([output_dim, output_dim_vec], dim=-1)
    hidden = lat.mean(dim=-2)
    out_scalar = out_scalar.mean(dim=-2)
    out_vec = torch.einsum("boc,od->bcd", out_vec, grid) / num_ori
    out = out_vec * out_scalar.unsqueeze(-1)
    if dim == 2:
        out = torch.cat([out, torch.zeros_like(out[..., :1])], dim=-1)
    return out.reshape(-1, out.shape[-1]), hidden.reshape(-1, hidden.shape[-1])


# --------------------------------------------------------------------------- EMPN (PonitaGCN)
def ponita_layer(x, kernel_basis, fiber_basis, edge_index, P, prefix, calibrate=False):
    """ponita.py:149-185 + 219-230 (SeparableFiberBundleConvNext, depthwise, layer_scale=None)."""
    msg = x[edge_index[0]] * F.linear(kernel_basis, P[f"{prefix}.conv.kernel.weight"])  # ponita.py:153
    x_1 = scatter_sum(msg, edge_index[1].long(), x.shape[0])  # ponita.py:161
    fk = F.linear(fiber_basis, P[f"{prefix}.conv.fiber_kernel.weight"])  # ponita.py:164
    x_2 = torch.einsum("boc,poc->bpc", x_1, fk) / fk.shape[-2]  # ponita.py:166
    if calibrate:  # ponita.py:178-180,187-192 (the call that calibrates still finishes with the old x_2)
        with torch.no_grad():
            s_in, s_1, s_2 = x.std(), x_1.std(), x_2.std()
            P[f"{prefix}.conv.kernel.weight"] = P[f"{prefix}.conv.kernel.weight"] * s_in / s_1
            P[f"{prefix}.conv.fiber_kernel.weight"] = P[f"{prefix}.conv.fiber_kernel.weight"] * s_1 / s_2
    h = x_2 + P[f"{prefix}.conv.bias"]
    h = F.layer_norm(h, (h.shape[-1],), P[f"{prefix}.norm.weight"], P[f"{prefix}.norm.bias"], 1e-5)
    h = F.gelu(F.linear(h, P[f"{prefix}.linear_1.weight"], P[f"{prefix}.linear_1.bias"]))
    h = F.linear(h, P[f"{prefix}.linear_2.weight"], P[f"{prefix}.linear_2.bias"])
    return h + x


def ponita_forward(P, x, pos, edge_index, num_layers, prefix="ponita", calibrate=False):
    """ponita.py:349-369 Ponita.forward (no last_feature_conditioning, no attention)."""
    grid = P[f"{prefix}.ori_grid"]
    kb = basis_mlp(spatial_invariants(grid, pos[edge_index[0]], pos[edge_index[1]]), P, f"{prefix}.basis_fn")
    fb = basis_mlp(orientation_invariants(grid), P, f"{prefix}.fiber_basis_fn")
    x = F.linear(x, P[f"{prefix}.x_embedder.weight"])
    for i in range(num_layers):
        x = ponita_layer(x, kb, fb, edge_index, P, f"{prefix}.interaction_layers.{i}", calibrate)
    return x


def homogeneous_graph(graph: dict, batch_size: int):
    """ponita_gcn.py:73-83 + PyG to_homogeneous [upstream]: per sample, node types are concatenated
    in graph["node_types"] order; edges of all types are merged with per-type node offsets.
    Returns (edge_index [2,E] over B*n_all nodes, n_all, per-type offsets)."""
    n_per = {t: graph["pos"][t].shape[0] // batch_size for t in graph["node_types"]}
    off, acc = {}, 0
    for t in graph["node_types"]:
        off[t] = acc
        acc += n_per[t]
    n_all = acc
    eis = []
    for (src, _, dst), ei in graph["edge_index"].items():
        if ei.numel() == 0:
            continue
        bs, ls = ei[0] // n_per[src], ei[0] % n_per[src]
        bd, ld = ei[1] // n_per[dst], ei[1] % n_per[dst]
        eis.append(torch.stack([bs * n_all + off[src] + ls, bd * n_all + off[dst] + ld]))
    return torch.cat(eis, dim=1), n_all, off, n_per


def empn_forward(P, graph, scalar_dict, vector_dict, *, dim, output_dim, output_dim_vec, num_layers, batch_size,
                 calibrate=False):
    """ponita_gcn.py:88-146 PonitaGCN.one_step."""
    grid = P["ponita.ori_grid"]
    xs, ps = [], []
    for t in graph["node_types"]:  # ponita_gcn.py:102-116
        x = lift_features(scalar_dict[t], vector_dict[t], grid, dim)
        xs.append(x.reshape(batch_size, -1, *x.shape[1:]))
        p = graph["pos"][t].reshape(batch_size, -1, 3)
        ps.append(p[..., :2] if dim == 2 else p)
    x = torch.cat(xs, dim=1)
    pos = torch.cat(ps, dim=1)
    ei, n_all, off, n_per = homogeneous_graph(graph, batch_size)
    x = x.reshape(-1, *x.shape[2:])
    pos = pos.reshape(-1, pos.shape[2])
    hidden = ponita_forward(P, x, pos, ei, num_layers, "ponita", calibrate)
    hidden = hidden.reshape(batch_size, -1, *hidden.shape[1:])
    k = graph["output_mask_key"]
    lat = hidden[:, off[k] : off[k] + n_per[k]]  # ponita_gcn.py:138-141 (mask commutes with the linear ops)
    lat = lat.reshape(-1, *lat.shape[2:])
    return readout(lat, P["linear.weight"], P["linear.bias"], grid, dim, output_dim, output_dim_vec)


# --------------------------------------------------------------------------- parameter init
def init_linear(out_f, in_f, bias=True, gen=None, dtype=torch.float32):
    """torch.nn.Linear default init (kaiming_uniform a=sqrt(5)) with an explicit generator."""
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen, dtype=dtype) * 2 - 1) * bound
    if not bias:
        return w, None
    b = (torch.rand(out_f, generator=gen, dtype=dtype) * 2 - 1) * bound
    return w, b


def init_conv_params(P, prefix, C=64, widen=4, gen=None, hepi=True):
    names = (
        dict(k="kernel", fk="fiber_kernel", b="bias", ln="node_mlp.0", l1="node_mlp.1", l2="node_mlp.3")
        if hepi
        else dict(k="conv.kernel", fk="conv.fiber_kernel", b="conv.bias", ln="norm", l1="linear_1", l2="linear_2")
    )
    P[f"{prefix}.{names['k']}.weight"], _ = init_linear(C, C, False, gen)
    P[f"{prefix}.{names['fk']}.weight"], _ = init_linear(C, C, False, gen)
    P[f"{prefix}.{names['b']}"] = torch.zeros(C)
    P[f"{prefix}.{names['ln']}.weight"] = torch.ones(C)
    P[f"{prefix}.{names['ln']}.bias"] = torch.zeros(C)
    P[f"{prefix}.{names['l1']}.weight"], P[f"{prefix}.{names['l1']}.bias"] = init_linear(C * widen, C, True, gen)
    P[f"{prefix}.{names['l2']}.weight"], P[f"{prefix}.{names['l2']}.bias"] = init_linear(C, C * widen, True, gen)


def init_hepi_params(input_dim_node, rounds, *, dim=3, num_ori=16, only_upper_hemisphere=False, C=64,
                     output_dim=1, output_dim_vec=1, seed=0) -> Dict[str, torch.Tensor]:
    """Parameter inventory of SURVEY Appendix B (hepi.py:61-107, conv.py:43-69)."""
    gen = torch.Generator().manual_seed(seed)
    P = {"ori_grid": make_grid(dim, num_ori, only_upper_hemisphere)}
    P["basis_fn.1.weight"], P["basis_fn.1.bias"] = init_linear(C, 14, True, gen)
    P["basis_fn.3.weight"], P["basis_fn.3.bias"] = init_linear(C, C, True, gen)
    P["fiber_basis_fn.1.weight"], P["fiber_basis_fn.1.bias"] = init_linear(C, 3, True, gen)
    P["fiber_basis_fn.3.weight"], P["fiber_basis_fn.3.bias"] = init_linear(C, C, True, gen)
    P["node_encoder.weight"], _ = init_linear(C, input_dim_node, False, gen)
    for k, r in enumerate(rounds):
        for et in r:
            init_conv_params(P, f"processor.{k}.convs.{conv_key(et)}", C, 4, gen, hepi=True)
    P["decoder.weight"], P["decoder.bias"] = init_linear(output_dim + output_dim_vec, C, True, gen)
    return P


def init_empn_params(input_dim_node, *, dim=3, num_ori=16, only_upper_hemisphere=False, C=64, num_layers=2,
                     output_dim=1, output_dim_vec=1, seed=0) -> Dict[str, torch.Tensor]:
    """ponita.py:276-325 + ponita_gcn.py:56 parameter inventory."""
    gen = torch.Generator().manual_seed(seed)
    P = {"ponita.ori_grid": make_grid(dim, num_ori, only_upper_hemisphere)}
    P["ponita.basis_fn.1.weight"], P["ponita.basis_fn.1.bias"] = init_linear(C, 14, True, gen)
    P["ponita.basis_fn.3.weight"], P["ponita.basis_fn.3.bias"] = init_linear(C, C, True, gen)
    P["ponita.fiber_basis_fn.1.weight"], P["ponita.fiber_basis_fn.1.bias"] = init_linear(C, 3, True, gen)
    P["ponita.fiber_basis_fn.3.weight"], P["ponita.fiber_basis_fn.3.bias"] = init_linear(C, C, True, gen)
    P["ponita.x_embedder.weight"], _ = init_linear(C, input_dim_node, False, gen)
    for i in range(num_layers):
        init_conv_params(P, f"ponita.interaction_layers.{i}", C, 4, gen, hepi=False)
    # the reference also owns an unused read_out layer (ponita.py:321-322); kept for state_dict parity
    P[f"ponita.read_out_layers.{num_layers - 1}.weight"], P[f"ponita.read_out_layers.{num_layers - 1}.bias"] = \
        init_linear(output_dim + output_dim_vec, C, True, gen)
    P["linear.weight"], P["linear.bias"] = init_linear(output_dim + output_dim_vec, C, True, gen)
    return P
