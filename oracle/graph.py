"""Oracle: observation split, batched hetero-graph topology, node features, DeepSets critic (plain torch, CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Abbreviations (geometry_rl/modules/): rigid.py = pyg_data/rigid_tasks_data.py, cloth.py = pyg_data/cloth_tasks_data.py,
rope.py = pyg_data/rope_tasks_data.py, tf.py = pyg_data/transforms.py, deepsets.py = pyg_models/deepsets.py,
vf.py = ../algorithms/trust_region_projections/models/value/gnn_vf_net.py.

PyG pieces (HeteroData/Batch.from_data_list/coalesce/node_type_subgraph, torch_cluster.knn_graph, nn.MLP,
LayerNorm(mode="graph")) are not importable here: restated from the call sites [upstream; PARITY UNPINNED].
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

EdgeType = Tuple[str, str, str]


@dataclass
class TaskSpec:
    """Static description of one task family's observation layout and graph schema."""
    family: str  # "rigid" | "cloth" | "rope"
    node_types: List[str]  # full NodeType enum order (one-hot width, tf.py:52-66)
    edge_types: List[EdgeType]  # EdgeType enum order
    edge_levels: List[str]
    obs_names: Dict[str, List[str]]  # group -> term names
    obs_dims: Dict[str, List[int]]  # group -> term widths
    num_actuators: int
    knn_k: int = 3
    angular_velocity: bool = True  # rigid only (rigid.py:65)
    n_vec: int = 4  # vectors per node: rigid 4 (rigid.py:219), cloth/rope 3
    in_features: List[str] = field(default_factory=list)


def rigid_spec(P=32, G=1, E_mesh=180, angular_velocity=True, object_velocity=True) -> TaskSpec:
    """orbit/tasks/manipulation/rigid_tasks/config/common_cfg/observations_cfg.py:143-192 (G=1) and :254-273 (two agents)."""
    vel_names = ["grippers"] + (["grippers_angular"] if angular_velocity else [])
    if object_velocity:
        vel_names += ["object_geometry"] + (["object_geometry_angular"] if angular_velocity else [])
    vel_dims = [3 * G if n.startswith("grippers") else 3 for n in vel_names]
    return TaskSpec(
        family="rigid",
        node_types=["object_geometry", "grippers", "target_geometry"],  # rigid.py:21-24
        edge_types=[("object_geometry", "internal", "object_geometry"), ("grippers", "agent", "grippers"),
                    ("object_geometry", "task", "grippers")],  # rigid.py:33-48
        edge_levels=["internal", "task", "agent"],  # rigid.py:27-30
        obs_names={"scalars": ["object_target_distances"],
                   "position_vectors": ["grippers", "object_geometry", "target_geometry"],
                   "velocity_vectors": vel_names,
                   "infos": ["object_num_points", "object_geometry_edges", "object_num_edges"]},
        obs_dims={"scalars": [1], "position_vectors": [3 * G, 3 * P, 3 * P], "velocity_vectors": vel_dims,
                  "infos": [1, 2 * E_mesh, 1]},
        num_actuators=G, angular_velocity=angular_velocity, n_vec=4,
        in_features=["scalars", "position_vectors", "velocity_vectors", "norm_position_vectors",
                     "norm_velocity_vectors", "infos"],
    )


def cloth_spec(n_particles=225, n_hole=10, G=4, E_cloth=600) -> TaskSpec:
    """orbit/tasks/manipulation/cloth_tasks/config/common_cfg/observations_cfg.py:150-193."""
    return TaskSpec(
        family="cloth",
        node_types=["particles", "grippers", "hole_boundary", "target_hook"],  # cloth.py:19-23
        edge_types=[("hole_boundary", "internal", "hole_boundary"), ("grippers", "agent", "grippers"),
                    ("hole_boundary", "task", "grippers")],  # cloth.py:32-47
        edge_levels=["internal", "task", "agent"],
        obs_names={"scalars": ["hole_target_distances", "cloth_edges_length"],
                   "position_vectors": ["grippers", "particles", "init_particles", "hole_boundary", "target_hook"],
                   "velocity_vectors": ["grippers", "particles"]},
        obs_dims={"scalars": [n_hole, E_cloth],
                  "position_vectors": [3 * G, 3 * n_particles, 3 * n_particles, 3 * n_hole, 3],
                  "velocity_vectors": [3 * G, 3 * n_particles]},
        num_actuators=G, n_vec=3,
        in_features=["scalars", "position_vectors", "velocity_vectors", "norm_position_vectors",
                     "norm_velocity_vectors"],
    )


def rope_spec(n_links=80, G=2, variable_length=False) -> TaskSpec:
    """orbit/tasks/manipulation/rope_tasks/config/common_cfg/observations_cfg.py:131-160.

    ``variable_length`` (BASELINE config 5 "variable-length rope graphs"; NOT in the reference, which asserts equal rope sizes within
    a batch, rope.py:127): an extra ``infos`` group carries ``links_num_points`` [B,1]; links / target points beyond it are zero
    padding exactly like the rigid tasks' padded object points (orbit/tasks/common/utils.py:193-211): no edges touch them."""
    spec = _rope_spec(n_links, G)
    if variable_length:
        spec.obs_names["infos"] = ["links_num_points"]
        spec.obs_dims["infos"] = [1]
        spec.in_features = spec.in_features + ["infos"]
    return spec


def _rope_spec(n_links=80, G=2) -> TaskSpec:
    return TaskSpec(
        family="rope",
        node_types=["links", "grippers", "target_geometry"],  # rope.py:21-24
        edge_types=[("links", "internal", "links"), ("grippers", "agent", "grippers"), ("links", "task", "grippers")],
        edge_levels=["internal", "task", "agent"],
        obs_names={"scalars": ["links_target_distances"],
                   "position_vectors": ["grippers", "links", "target_geometry"],
                   "velocity_vectors": ["grippers", "links"]},
        obs_dims={"scalars": [1], "position_vectors": [3 * G, 3 * n_links, 3 * n_links],
                  "velocity_vectors": [3 * G, 3 * n_links]},
        num_actuators=G, n_vec=3,
        in_features=["scalars", "position_vectors", "velocity_vectors", "norm_position_vectors",
                     "norm_velocity_vectors"],
    )


def kept_node_types(spec: TaskSpec, full_graph_obs: bool) -> List[str]:
    """rigid.py:91, cloth.py:87-91, rope.py:89."""
    if spec.family == "rigid":
        return [t for t in spec.node_types if t != "target_geometry"]
    if spec.family == "cloth":
        keep = [t for t in spec.node_types if t != "target_hook"]
        return keep if full_graph_obs else [t for t in keep if t != "particles"]
    return list(spec.node_types)


def split_obs(spec: TaskSpec, obs: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """rigid.py:93-150 _preprocess_input (cloth.py:93-142, rope.py:91-141): torch.split by term widths; vectors -> [B,n,3]."""
    B = obs["scalars"].shape[0]
    out = {}
    for group in ["scalars", "position_vectors", "velocity_vectors", "norm_position_vectors",
                  "norm_velocity_vectors", "infos"]:
        if group not in obs:
            continue
        base = group.replace("norm_", "")
        parts = torch.split(obs[group], spec.obs_dims[base], dim=1)
        d = {}
        for name, part in zip(spec.obs_names[base], parts):
            d[name] = part.reshape(B, -1, 3) if "vectors" in group else part
        out[group] = d
    return out


def knn_edges(points: torch.Tensor, k: int) -> torch.Tensor:
    """torch_cluster.knn_graph(x, k) [upstream]: for every node its k nearest OTHER nodes as sources
    (flow source_to_target): edge_index = [neighbour, centre].  Ties unpinned."""
    n = points.shape[0]
    if n <= 1:
        return torch.zeros(2, 0, dtype=torch.long)
    d = torch.cdist(points.double(), points.double())
    d.fill_diagonal_(float("inf"))
    kk = min(k, n - 1)
    nbr = d.topk(kk, dim=1, largest=False).indices  # [n, kk]
    centre = torch.arange(n)[:, None].expand(n, kk)
    return torch.stack([nbr.reshape(-1), centre.reshape(-1)])


def full_edges(n_src: int, n_dst: int, exclude_self: bool) -> torch.Tensor:
    """rigid.py:289-300,313-319 (and the cloth/rope equivalents): j -> k double loops."""
    j, k = torch.meshgrid(torch.arange(n_src), torch.arange(n_dst), indexing="ij")
    j, k = j.reshape(-1), k.reshape(-1)
    if exclude_self:
        m = j != k
        j, k = j[m], k[m]
    return torch.stack([j, k])


def build_topology(spec: TaskSpec, split: dict, full_graph_obs: bool) -> dict:
    """_construct_placeholders (rigid.py:257-343, cloth.py:224-307, rope.py:227-300) without PyG:
    batched edge_index per edge type with per-sample node offsets, sorted by (dst, src) (coalesce() only fixes a
    canonical order; sums are order-independent up to rounding)."""
    posv = split["position_vectors"]
    B = posv["grippers"].shape[0]
    keep = kept_node_types(spec, full_graph_obs)
    n_per = {t: posv[t].shape[1] for t in spec.node_types}
    edges: Dict[EdgeType, List[torch.Tensor]] = {et: [] for et in spec.edge_types}
    G = n_per["grippers"]
    for i in range(B):
        if spec.family == "rigid":
            p = int(split["infos"]["object_num_points"][i].long().item())
            pts = posv["object_geometry"][i][:p]
            loc = {spec.edge_types[0]: knn_edges(pts, spec.knn_k),  # rigid.py:285-287
                   spec.edge_types[1]: full_edges(G, G, True) if G > 1 else torch.zeros(2, 0, dtype=torch.long),
                   spec.edge_types[2]: full_edges(p, G, False)}  # rigid.py:313-319
        elif spec.family == "cloth":
            H = n_per["hole_boundary"]
            loc = {spec.edge_types[0]: full_edges(H, H, True), spec.edge_types[1]: full_edges(G, G, True),
                   spec.edge_types[2]: full_edges(H, G, False)}
        else:
            L = n_per["links"]
            if "infos" in split and "links_num_points" in split["infos"]:   # variable-length ropes: only the valid links have edges
                L = min(L, int(split["infos"]["links_num_points"][i].long().item()))
            loc = {spec.edge_types[0]: knn_edges(posv["links"][i][:L], spec.knn_k),  # rope.py:251
                   spec.edge_types[1]: full_edges(G, G, True), spec.edge_types[2]: full_edges(L, G, False)}
        for et, ei in loc.items():
            src, _, dst = et
            off = torch.tensor([[i * n_per[src]], [i * n_per[dst]]])
            edges[et].append(ei + off)
    edge_index = {}
    for et, lst in edges.items():
        if et[0] not in keep or et[2] not in keep:
            continue
        ei = torch.cat(lst, dim=1)
        order = torch.argsort(ei[1] * (ei[0].max() + 1 if ei.numel() else 1) + ei[0])
        edge_index[et] = ei[:, order]
    return {"node_types": keep, "edge_index": edge_index, "n_per": {t: n_per[t] for t in keep}, "batch_size": B,
            "all_node_types": list(spec.node_types)}


def build_features(spec: TaskSpec, topo: dict, split: dict, dist_as_pos: bool):
    """_update_placeholders + construct_input_vector (rigid.py:152-252, cloth.py:144-222, rope.py:143-225)
    with training_noise=False.  Returns (graph, scalar_dict, vector_dict)."""
    npos, nvel, pos = split["norm_position_vectors"], split["norm_velocity_vectors"], split["position_vectors"]
    graph = {"node_types": topo["node_types"], "edge_index": topo["edge_index"], "pos": {}, "batch_size": topo["batch_size"]}
    scalar_dict, vector_dict = {}, {}
    n_types = len(spec.node_types)
    for t in topo["node_types"]:
        graph["pos"][t] = pos[t].reshape(-1, 3)
        norm_pos = npos[t].reshape(-1, 3)
        one_hot = torch.zeros(norm_pos.shape[0], n_types, dtype=norm_pos.dtype)
        one_hot[:, spec.node_types.index(t)] = 1  # tf.py:52-66 (index among ALL node types)
        zeros = torch.zeros_like(norm_pos)
        if spec.family == "rigid":
            if t == "object_geometry":  # rigid.py:185-192
                target = npos["target_geometry"].reshape(-1, 3)
                corr = norm_pos - target if dist_as_pos else target
            else:
                corr = zeros
            if t in nvel:  # rigid.py:194-218
                rep = npos[t].shape[1] if t == "object_geometry" else 1
                vel = nvel[t].repeat_interleave(rep, dim=1).reshape(-1, 3) if rep > 1 else nvel[t].reshape(-1, 3)
                if spec.angular_velocity:
                    a = nvel[f"{t}_angular"]
                    ang = a.repeat_interleave(rep, dim=1).reshape(-1, 3) if rep > 1 else a.reshape(-1, 3)
                else:
                    ang = torch.zeros_like(vel)
            else:
                vel, ang = zeros, zeros
            vectors = torch.cat([norm_pos, corr, vel, ang], dim=1)
        elif spec.family == "cloth":
            if t == "particles":  # cloth.py:173-175
                init = npos["init_particles"].reshape(-1, 3)
                corr = norm_pos - init if dist_as_pos else init
            elif t == "hole_boundary":  # cloth.py:176-181
                nh = npos["hole_boundary"].shape[1]
                target = torch.repeat_interleave(npos["target_hook"], nh, 1).reshape(-1, 3)
                corr = norm_pos - target if dist_as_pos else target
            else:
                corr = zeros
            vel = nvel[t].reshape(-1, 3) if t in nvel else zeros
            vectors = torch.cat([norm_pos, corr, vel], dim=1)
        else:
            if t == "links":  # rope.py:172-178
                target = npos["target_geometry"].reshape(-1, 3)
                corr = norm_pos - target if dist_as_pos else target
            else:
                corr = zeros
            vel = nvel[t].reshape(-1, 3) if t in nvel else zeros
            vectors = torch.cat([norm_pos, corr, vel], dim=1)
        scalar_dict[t], vector_dict[t] = one_hot, vectors
    return graph, scalar_dict, vector_dict


# --------------------------------------------------------------------------- DeepSets critic
def graph_layer_norm(x, w, b, eps=1e-5, stats=None):
    """PyG 2.5.2 LayerNorm(mode="graph") with batch=None [upstream]: statistics over ALL elements, biased std,
    eps added to the std.  ``stats`` = (mean, std) overrides (data-parallel check)."""
    if stats is None:
        mean = x.mean()
        std = (x - mean).std(unbiased=False)
    else:
        mean, std = stats
    return (x - mean) / (std + eps) * w + b


def deepsets_forward(P: Dict[str, torch.Tensor], x: torch.Tensor, prefix="gnn", stats_fn=None):
    """deepsets.py:34-53 + PyG MLP([in,64,64], norm="layer_norm") [upstream]: Linear, graph-LN, ReLU, Linear; sum over nodes;
    second MLP.  x [B, n_all, d] -> [B, 64]."""
    h = F.linear(x, P[f"{prefix}.mlp_inner.lins.0.weight"], P[f"{prefix}.mlp_inner.lins.0.bias"])
    st = stats_fn(h) if stats_fn else None  # data-parallel check: statistics of the GLOBAL batch
    h = F.relu(graph_layer_norm(h, P[f"{prefix}.mlp_inner.norms.0.weight"], P[f"{prefix}.mlp_inner.norms.0.bias"], stats=st))
    h = F.linear(h, P[f"{prefix}.mlp_inner.lins.1.weight"], P[f"{prefix}.mlp_inner.lins.1.bias"])
    z = h.sum(dim=1)  # deepsets.py:51
    u = F.linear(z, P[f"{prefix}.mlp_outer.lins.0.weight"], P[f"{prefix}.mlp_outer.lins.0.bias"])
    st = stats_fn(u) if stats_fn else None
    u = F.relu(graph_layer_norm(u, P[f"{prefix}.mlp_outer.norms.0.weight"], P[f"{prefix}.mlp_outer.norms.0.bias"], stats=st))
    return F.linear(u, P[f"{prefix}.mlp_outer.lins.1.weight"], P[f"{prefix}.mlp_outer.lins.1.bias"])


def critic_input(topo: dict, scalar_dict, vector_dict) -> torch.Tensor:
    """deepsets.py:41-49 with concat_input_vector=True (rigid.py:220-228): [B, n_all, n_types + 3 n_vec]."""
    B = topo["batch_size"]
    xs = []
    for t in topo["node_types"]:
        u = torch.cat([scalar_dict[t], vector_dict[t]], dim=1)
        xs.append(u.reshape(B, -1, u.shape[-1]))
    return torch.cat(xs, dim=1)


def value_forward(P: Dict[str, torch.Tensor], x: torch.Tensor, stats_fn=None) -> torch.Tensor:
    """vf.py:50-86 GNNVFNet.forward for a 2-D batch: DeepSets -> Linear(64,1).  [B, n_all, d] -> [B, 1]."""
    return F.linear(deepsets_forward(P, x, "gnn", stats_fn), P["final.weight"], P["final.bias"])


def init_critic_params(in_dim: int, hidden=64, seed=1) -> Dict[str, torch.Tensor]:
    """deepsets.py:22-23 (PyG Linear default init == torch default) + vf.py:48; builders/utils_algo_graph.py:195-198
    re-initialises only torch.nn.Linear modules (the final layer) with orthogonal(0.01), zero bias."""
    from .equivariant import init_linear
    gen = torch.Generator().manual_seed(seed)
    P = {}
    for name, (o, i) in {"mlp_inner.lins.0": (hidden, in_dim), "mlp_inner.lins.1": (hidden, hidden),
                         "mlp_outer.lins.0": (hidden, hidden), "mlp_outer.lins.1": (hidden, hidden)}.items():
        P[f"gnn.{name}.weight"], P[f"gnn.{name}.bias"] = init_linear(o, i, True, gen)
    for name in ["mlp_inner.norms.0", "mlp_outer.norms.0"]:
        P[f"gnn.{name}.weight"], P[f"gnn.{name}.bias"] = torch.ones(hidden), torch.zeros(hidden)
    w = torch.empty(1, hidden)
    torch.nn.init.orthogonal_(w, 0.01, generator=gen)
    P["final.weight"], P["final.bias"] = w, torch.zeros(1)
    return P
