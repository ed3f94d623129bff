#!/usr/bin/env python3
"""Generate golden vectors for the oracle by importing the reference IN THIS CONTAINER.

Run:  python tools/make_golden.py            (needs /root/reference; writes tests/golden/*.npz)

The reference cannot be imported end-to-end here (PyG, torch_scatter, torchrl, tensordict, ITPAL are absent, SURVEY.md
section 8c), so three tiers are used:

  tier 1  modules that import unmodified (ponita.py, to_from_sphere.py, torch_utils.py): grids, polynomial features,
          the full Ponita (EMPN core) forward/backward, calibration.
  tier 2  modules imported under NAME-ONLY stubs for gymnasium / stable_baselines3 (no arithmetic in the stubs):
          gaussian_kl, mean_projection, get_trust_region_loss, compute_metrics, GNNGaussianPolicyDiag helpers + std head.
  tier 2b HEPi.one_step / FiberBundleConv / HeteroFiberConv run as reference code under stubs of the PyG container
          classes.  These stubs DO carry the three PyG semantics the call sites rely on -- gather x_src[edge_index[0]],
          scatter-sum over edge_index[1], sum of per-edge-type outputs -- restated from PyG 2.5.2; fixtures from this
          tier are flagged ``tier2b`` and pin everything else in those files (invariants, bases, message, einsum,
          calibration order, node MLP, readout).

Only inputs/outputs (arrays) are written; no reference source text is stored.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(OUT.rstrip("/")).rsplit("/tests", 1)[0])


def npd(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


# ----------------------------------------------------------------------------------------------- stubs
def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Env:  # gymnasium.Env (name only)
        pass

    mod("gymnasium", Env=_Env)
    mod("stable_baselines3")
    mod("stable_baselines3.common")
    mod("stable_baselines3.common.vec_env", VecEnvWrapper=object, VecNormalize=object)

    # --- PyG containers (tier 2b) -------------------------------------------------------------
    class AttentionalAggregation(nn.Module):
        """PyG 2.5.2 nn/aggr/attention.py (gate_nn only) + utils.softmax, restated with dense one-hot products so that the reference's
        torch.vmap over the orientation axis (conv.py:58-61) can trace it: gate = gate_nn(x); alpha = exp(gate - max_group) /
        (sum_group + 1e-16), max detached; out = sum_group(alpha * x)."""

        def __init__(self, gate_nn, nn=None):
            super().__init__()
            self.gate_nn = gate_nn

        def forward(self, x, index, ptr=None, dim_size=None, dim=-2):
            gate = self.gate_nn(x)
            M = torch.nn.functional.one_hot(index, dim_size).to(x.dtype)                   # [E, Nd]
            mx = (gate.detach()[:, None, :] + (M[:, :, None] - 1.0) * 1e30).amax(0)        # [Nd, C]
            ex = (gate - M @ mx).exp()
            den = M.t() @ ex + 1e-16
            return M.t() @ (ex / (M @ den) * x)

    class MessagePassing(nn.Module):
        def __init__(self, node_dim=0, aggr="add", aggr_kwargs=None, **kw):
            super().__init__()
            if aggr == "AttentionalAggregation":
                self.aggr_module = AttentionalAggregation(**(aggr_kwargs or {}))

        def propagate(self, edge_index, size=None, x=None, kernel=None, dim_size=None, **kw):
            x_src, x_dst = x
            x_j = x_src[edge_index[0]]
            x_i = x_dst[edge_index[1]]
            msg = self.message(x_i=x_i, x_j=x_j, kernel=kernel)
            return self.aggregate(msg, edge_index, dim_size=dim_size)

    def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
        assert reduce in ("sum", "add") and dim == 0
        out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
        return out.index_add(0, index, src)

    class _TupleModuleDict(nn.ModuleDict):
        def __init__(self, modules):
            super().__init__()
            self._keys_orig = {}
            for k, v in modules.items():
                ik = "<" + "___".join(k) + ">"
                self._keys_orig[ik] = tuple(k)
                self[ik] = v

        def items(self):
            return [(self._keys_orig[k], v) for k, v in super().items()]

    class HeteroConv(nn.Module):
        def __init__(self, convs, aggr="sum"):
            super().__init__()
            self.convs = _TupleModuleDict(convs)
            self.aggr = aggr

    def group(xs, aggr):
        assert aggr == "sum"
        return xs[0] if len(xs) == 1 else torch.stack(xs, 0).sum(0)

    class Data:
        pass

    class HeteroData:
        pass

    tg = mod("torch_geometric")
    tg.nn = mod("torch_geometric.nn", MessagePassing=MessagePassing, MLP=object)
    tg.data = mod("torch_geometric.data", Data=Data, HeteroData=HeteroData)
    conv = mod("torch_geometric.nn.conv", MessagePassing=MessagePassing, HeteroConv=HeteroConv)
    mod("torch_geometric.nn.conv.hetero_conv", group=group)
    mod("torch_geometric.typing", EdgeType=tuple, NodeType=str)
    mod("torch_scatter", scatter=scatter)
    tg.nn.conv = conv


# ----------------------------------------------------------------------------------------------- tier 1
def tier1():
    from geometry_rl.modules.pyg_models.ponita.ponita import GridGenerator, PolynomialFeatures, Ponita
    from geometry_rl.algorithms.trust_region_projections.utils.torch_utils import inverse_softplus

    out = {}
    out["grid_s1_16"] = GridGenerator(2, 16)()
    out["grid_s2_16"] = GridGenerator(3, 16)()
    out["grid_s2_16_upper"] = GridGenerator(3, 16, only_upper_hemisphere=True)()
    out["grid_s2_20"] = GridGenerator(3, 20)()
    g = torch.Generator().manual_seed(1)
    x2 = torch.randn(5, 16, 2, generator=g)
    x1 = torch.randn(4, 4, 1, generator=g)
    out["poly_in2"], out["poly_out2"] = x2, PolynomialFeatures(2)(x2)
    out["poly_in1"], out["poly_out1"] = x1, PolynomialFeatures(2)(x1)
    xs = torch.tensor([0.3, 1.0, 2.5])
    out["inv_softplus_in"], out["inv_softplus_out"] = xs, inverse_softplus(xs)
    np.savez(os.path.join(OUT, "tier1_basics.npz"), **npd(out))

    for dim in (3, 2):
        torch.manual_seed(10 + dim)
        net = Ponita(input_dim=7, hidden_dim=64, output_dim=1, num_layers=2, output_dim_vec=1, dim=dim, num_ori=16,
                     degree=2, widening_factor=4, layer_scale=None, task_level="node")
        N, E = 14, 40
        x = torch.randn(N, 16, 7, generator=g)
        pos = torch.randn(N, dim, generator=g)
        ei = torch.stack([torch.randint(0, N, (E,), generator=g), torch.randint(0, N, (E,), generator=g)])
        R = torch.randn(N, 16, 64, generator=g)
        rec = {"x": x, "pos": pos, "edge_index": ei, "R": R}
        rec.update({"init." + k: v.clone() for k, v in net.state_dict().items()})
        net.train()
        y0 = net(x, pos, ei)  # first training call: calibrates (ponita.py:178-180)
        rec["y_first_call"] = y0
        rec.update({"cal." + k: v.clone() for k, v in net.state_dict().items()})
        net.zero_grad()
        xg = x.clone().requires_grad_(True)
        y = net(xg, pos, ei)
        (y * R).sum().backward()
        rec["y"] = y
        rec["grad.x"] = xg.grad
        for k, p in net.named_parameters():
            if p.grad is not None:
                rec["grad." + k] = p.grad
        np.savez(os.path.join(OUT, f"tier1_ponita_dim{dim}.npz"), **npd(rec))


# ----------------------------------------------------------------------------------------------- tier 2
def tier2():
    from geometry_rl.algorithms.trust_region_projections.projections.base_projection_layer import (
        BaseProjectionLayer, mean_projection)
    from geometry_rl.algorithms.trust_region_projections.utils.projection_utils import gaussian_kl
    from geometry_rl.algorithms.trust_region_projections.models.policy.gnn_gaussian_policy_diag import (
        GNNGaussianPolicyDiag)

    g = torch.Generator().manual_seed(5)
    B, A = 9, 6

    class FakeGNN(nn.Module):
        device = "cpu"

        def one_step(self, data, input_vector):
            return data

    class FakeData:
        def build_data(self, *args, train=True):
            return self.payload, None

    torch.manual_seed(3)
    fd = FakeData()
    policy = GNNGaussianPolicyDiag(gnn=FakeGNN(), hyper_data=fd, action_dim=A, num_actuators=1, init="orthogonal",
                                   hidden_sizes=(64, 64), contextual_std=True, init_std=1.0, minimal_std=1e-5,
                                   share_action_dim=True, post_fc=False)
    hidden = torch.randn(B, 64, generator=g)
    mean_in = torch.randn(B * 2, 3, generator=g)
    fd.payload = (mean_in, hidden)
    with torch.no_grad():
        policy._pre_std.weight.mul_(30.0)  # make the contextual std visibly state dependent
    loc, cov = policy(torch.zeros(B, 1), train=True)
    rec = {"hidden": hidden, "gnn_out": mean_in, "loc": loc, "cov": cov,
           "pre_std.weight": policy._pre_std.weight, "pre_std.bias": policy._pre_std.bias}

    mean = torch.randn(B, A, generator=g)
    S = (torch.rand(B, A, generator=g) + 0.5).diag_embed()
    mean_o = mean + 0.3 * torch.randn(B, A, generator=g)
    mean_o[0] = mean[0] + 1e-3  # one sample inside the mean bound
    S_o = (torch.rand(B, A, generator=g) + 0.5).diag_embed()
    x = torch.randn(B, A, generator=g)
    p, q = (mean, S), (mean_o, S_o)
    rec.update({"mean": mean, "S": S, "mean_o": mean_o, "S_o": S_o, "x": x})
    rec["maha"] = policy.maha(mean, mean_o, S_o)
    rec["logdet"] = policy.log_determinant(S)
    rec["entropy"] = policy.entropy(p)
    rec["log_prob"] = policy.log_probability(p, x)
    rec["covariance"] = policy.covariance(S)
    rec["precision"] = policy.precision(S)
    mp, cp = gaussian_kl(policy, p, q)
    rec["kl_mean"], rec["kl_cov"] = mp, cp
    eps = torch.tensor(0.05)
    rec["eps_mean"] = eps
    rec["proj_mean"] = mean_projection(mean, mean_o, mp, eps)
    rec["proj_mean_noop"] = mean_projection(mean, mean_o, mp, torch.tensor(1e6))
    layer = BaseProjectionLayer(proj_type="kl", mean_bound=0.05, cov_bound=0.0025, trust_region_coeff=4.0,
                                scale_prec=True, entropy_schedule=False, action_dim=A, total_train_steps=100,
                                cpu=True, dtype=torch.float32)
    proj = (rec["proj_mean"], (S * 0.9 + S_o * 0.1))
    mean_g = mean.clone().requires_grad_(True)
    S_g = S.clone().requires_grad_(True)
    trl = layer.get_trust_region_loss(policy, (mean_g, S_g), proj)
    trl.backward()
    rec["tr_proj_S"] = proj[1]
    rec["tr_loss"], rec["tr_grad_mean"], rec["tr_grad_S"] = trl, mean_g.grad, S_g.grad
    m = layer.compute_metrics(policy, p, proj, step=0)
    for k, v in m.items():
        rec["metric." + k] = v
    # base_projection_layer.__call__ with the base (identity) trust-region hook: entropy projection with bound -inf
    out_p = layer(policy, p, q, 0)
    rec["base_call_mean"], rec["base_call_S"] = out_p
    np.savez(os.path.join(OUT, "tier2_projection.npz"), **npd(rec))


# ----------------------------------------------------------------------------------------------- tier 2b
def tier2b(attention=False):
    """``attention``: one more case, FiberBundleConv(aggr="AttentionalAggregation") in every round (hepi_attention.yaml), written to
    tier2b_hepi_rigid_g2_attention.npz; the other fixtures are not touched."""
    from geometry_rl.modules.pyg_models.hepi import HEPi
    from geometry_rl.modules.pyg_models.ponita.conv import FiberBundleConv
    from oracle import graph as gr
    from geometry_rl_amd import synthetic as syn

    cases = {
        "rigid_g1": dict(spec=gr.rigid_spec(P=8, G=1, E_mesh=4), obs=lambda s: syn.make_rigid_obs(5, P=8, G=1, E_mesh=4, seed=s),
                         dim=3, upper=True, od=2, ov=2),
        "rigid_g2": dict(spec=gr.rigid_spec(P=8, G=2, E_mesh=4, angular_velocity=False, object_velocity=False),
                         obs=lambda s: syn.make_rigid_obs(4, P=8, G=2, E_mesh=4, angular_velocity=False,
                                                          object_velocity=False, seed=s),
                         dim=3, upper=False, od=1, ov=1),
        "rope_dim2": dict(spec=gr.rope_spec(n_links=7, G=2), obs=lambda s: syn.make_rope_obs(3, n_links=7, G=2, seed=s),
                          dim=2, upper=False, od=1, ov=1),
    }
    codes = [[1, 0], [0, 1], [0, 1]]
    aggr = "add"
    if attention:
        cases = {"rigid_g2_attention": cases["rigid_g2"]}
        aggr = "AttentionalAggregation"
    for name, c in cases.items():
        spec = c["spec"]
        obs = c["obs"](21)
        split = gr.split_obs(spec, obs)
        topo = gr.build_topology(spec, split, full_graph_obs=False)
        graph, s_dict, v_dict = gr.build_features(spec, topo, split, dist_as_pos=True)

        torch.manual_seed(77)
        mp = []
        for lvl in range(3):
            mp.append([FiberBundleConv(64, 64, 64, groups=64, separable=True, widening_factor=4, aggr=aggr) if codes[lvl][k] else None
                       for k in range(2)])
        n_in = len(spec.node_types) + spec.n_vec
        net = HEPi(input_dim_node=n_in, input_dim_edge=0, hidden_dim=64, latent_dim=64, output_dim=c["od"],
                   output_dim_vec=c["ov"], node_encoder_layers=2, edge_encoder_layers=2, node_decoder_layers=2,
                   node_type_mapping=None, edge_type_mapping=[tuple(e) for e in spec.edge_types],
                   edge_level_mapping=spec.edge_levels, message_passing=mp, num_messages=2, device="cpu", num_ori=16,
                   degree=2, ponita_dim=c["dim"], only_upper_hemisphere=c["upper"])

        class NS:
            pass

        class G:
            def __getitem__(self, k):
                n = NS()
                n.pos = graph["pos"][k]
                return n

        hg = G()
        hg.node_types = graph["node_types"]
        hg.edge_types = list(graph["edge_index"].keys())
        hg.edge_index_dict = graph["edge_index"]
        hg.output_mask_key = "grippers"

        rec = {"obs." + k: v for k, v in obs.items()}
        for et, ei in graph["edge_index"].items():
            rec["edge_index." + "|".join(et)] = ei
        rec.update({"init." + k: v.clone() for k, v in net.state_dict().items()})
        net.train()
        out0, hid0 = net.one_step(hg, (s_dict, v_dict))  # calibrating call (conv.py:104-105)
        rec["out_first_call"], rec["hidden_first_call"] = out0, hid0
        rec.update({"cal." + k: v.clone() for k, v in net.state_dict().items()})
        net.zero_grad()
        out, hid = net.one_step(hg, (s_dict, v_dict))
        g = torch.Generator().manual_seed(9)
        Ro, Rh = torch.randn(out.shape, generator=g), torch.randn(hid.shape, generator=g)
        ((out * Ro).sum() + (hid * Rh).sum()).backward()
        rec.update({"out": out, "hidden": hid, "R_out": Ro, "R_hidden": Rh})
        for k, p in net.named_parameters():
            if p.grad is not None:
                rec["grad." + k] = p.grad
        np.savez(os.path.join(OUT, f"tier2b_hepi_{name}.npz"), **npd(rec))


# ----------------------------------------------------------------------------------------------- tier 2c
def tier2c():
    """Frobenius and Wasserstein projection layers (frob_projection_layer.py:9-88, w2_projection_layer.py:14-76) on the diagonal
    policy: projection outputs, gradients through the projection, trust-region loss + gradients, metrics."""
    from geometry_rl.algorithms.trust_region_projections.projections.frob_projection_layer import FrobeniusProjectionLayer
    from geometry_rl.algorithms.trust_region_projections.projections.w2_projection_layer import WassersteinProjectionLayer
    from geometry_rl.algorithms.trust_region_projections.models.policy.gnn_gaussian_policy_diag import (
        GNNGaussianPolicyDiag)

    class FakeGNN(nn.Module):
        device = "cpu"

    class FakeData:
        pass

    torch.manual_seed(3)
    B, A = 11, 6
    policy = GNNGaussianPolicyDiag(gnn=FakeGNN(), hyper_data=FakeData(), action_dim=A, num_actuators=1, init="orthogonal",
                                   hidden_sizes=(64, 64), contextual_std=True, init_std=1.0, minimal_std=1e-5,
                                   share_action_dim=True, post_fc=False)
    for name, cls in (("frob", FrobeniusProjectionLayer), ("w2", WassersteinProjectionLayer)):
        g = torch.Generator().manual_seed(17)
        mean = torch.randn(B, A, generator=g)
        S = (torch.rand(B, A, generator=g) + 0.5)
        mean_o = mean + 0.3 * torch.randn(B, A, generator=g)
        S_o = (torch.rand(B, A, generator=g) + 0.5)
        mean_o[0] = mean[0] + 1e-3          # inside the mean bound
        S_o[1] = S[1] * (1 + 1e-3)          # inside the covariance bound
        mean_o[2] = mean[2] + 1e-3
        S_o[2] = S[2] * (1 - 1e-3)          # inside both
        R1, R2 = torch.randn(B, A, generator=g), torch.randn(B, A, generator=g)
        layer = cls(proj_type=name, mean_bound=0.05, cov_bound=0.0025, trust_region_coeff=4.0, scale_prec=True,
                    entropy_schedule=False, action_dim=A, total_train_steps=100, cpu=True, dtype=torch.float32)
        mean_g = mean.clone().requires_grad_(True)
        S_g = S.clone().requires_grad_(True)
        p = (mean_g, S_g.diag_embed())
        q = (mean_o, S_o.diag_embed())
        pm, pS = layer(policy, p, q, 0)
        rec = {"mean": mean, "S": S, "mean_o": mean_o, "S_o": S_o, "R1": R1, "R2": R2, "proj_mean": pm,
               "proj_S": pS.diagonal(dim1=-2, dim2=-1), "mean_bound": torch.tensor(0.05), "cov_bound": torch.tensor(0.0025),
               "coeff": torch.tensor(4.0)}
        ((pm * R1).sum() + (pS.diagonal(dim1=-2, dim2=-1) * R2).sum()).backward(retain_graph=True)
        rec["grad_mean"], rec["grad_S"] = mean_g.grad.clone(), S_g.grad.clone()
        mean_g.grad = None
        S_g.grad = None
        trl = layer.get_trust_region_loss(policy, p, (pm, pS))
        trl.backward()
        rec["tr_loss"], rec["tr_grad_mean"], rec["tr_grad_S"] = trl, mean_g.grad.clone(), S_g.grad.clone()
        m = layer.compute_metrics(policy, (mean, S.diag_embed()), (pm.detach(), pS.detach()), step=0)
        for k, v in m.items():
            rec["metric." + k] = v
        mp, cp = layer.trust_region_value(policy, (mean, S.diag_embed()), q)
        rec["value_mean"], rec["value_cov"] = mp, cp
        np.savez(os.path.join(OUT, f"tier2c_projection_{name}.npz"), **npd(rec))


def tier2d():
    """BASELINE config 1 (rigid_insertion_multi_transformer_trpl): the reference's TransformerVanilla (transformer_vanilla.py:10-92,
    configs/algorithm/pyg_agent/model/transformer.yaml) inside the reference's GNNGaussianPolicyDiag with post_fc=True
    (gnn_gaussian_policy_diag.py:26-87), forward and backward.  Stubs: PyG's ``MLP([64, 64], norm=None)`` -- one Linear layer stored as
    ``lins.0`` [upstream PyG 2.5.2: plain_last=True, no norm, no activation for a two-entry channel list] -- and a graph object carrying
    ``len()``, ``node_types`` and ``output_mask`` (what one_step reads, transformer_vanilla.py:59-66,88)."""
    import torch_geometric.nn as tgnn

    class MLP(nn.Module):   # PyG MLP restated for the only form the call site uses
        def __init__(self, channel_list, norm=None, **kw):
            super().__init__()
            assert len(channel_list) == 2 and norm is None
            self.lins = nn.ModuleList([nn.Linear(channel_list[0], channel_list[1])])

        def forward(self, x):
            return self.lins[0](x)

    tgnn.MLP = MLP
    sys.modules.pop("geometry_rl.modules.pyg_models.transformer_vanilla", None)
    from geometry_rl.modules.pyg_models.transformer_vanilla import TransformerVanilla
    from geometry_rl.algorithms.trust_region_projections.models.policy.gnn_gaussian_policy_diag import GNNGaussianPolicyDiag

    B, P, G, d, A = 5, 32, 1, 15, 6
    g = torch.Generator().manual_seed(17)

    class Graph:
        node_types = ["object_geometry", "grippers"]
        output_mask = slice(P, P + G)

        def __len__(self):
            return B

    class FakeData:
        def build_data(self, *args, train=True):
            return Graph(), self.payload

    torch.manual_seed(7)
    gnn = TransformerVanilla(input_dim_node=d, output_dim=64, num_layers=2, num_heads=2, hidden_dim=64, dropout=0.0, concat_global=False)
    fd = FakeData()
    policy = GNNGaussianPolicyDiag(gnn=gnn, hyper_data=fd, action_dim=A, num_actuators=G, init="orthogonal", hidden_sizes=(64, 64),
                                   contextual_std=True, init_std=1.0, minimal_std=1e-5, share_action_dim=True, post_fc=True)
    with torch.no_grad():   # the orthogonal(0.01) heads would hide the transformer behind a ~0 mean: make them visible
        policy._mean.weight.mul_(30.0)
        policy._pre_std.weight.mul_(30.0)
    u_obj = torch.randn(B * P, d, generator=g)
    u_grip = torch.randn(B * G, d, generator=g)
    fd.payload = {"object_geometry": u_obj, "grippers": u_grip}
    loc, cov = policy(torch.zeros(B, 1), train=True)
    w_loc, w_cov = torch.randn(loc.shape, generator=g), torch.randn(B, A, generator=g)
    (loc * w_loc).sum().add((cov.diagonal(dim1=-2, dim2=-1) * w_cov).sum()).backward()
    rec = {"u_object_geometry": u_obj, "u_grippers": u_grip, "loc": loc, "cov": cov, "w_loc": w_loc, "w_cov": w_cov,
           "B": np.int64(B), "P": np.int64(P), "G": np.int64(G)}
    for k, v in policy.state_dict().items():
        rec["param." + k] = v
    for k, p_ in policy.named_parameters():
        if p_.grad is not None:
            rec["grad." + k] = p_.grad
    np.savez(os.path.join(OUT, "tier2d_transformer_post_fc.npz"), **npd(rec))
    print("tier2d: ", len(rec), "arrays;", sum(p_.numel() for p_ in policy.parameters()), "parameters; loc", tuple(loc.shape))


# ----------------------------------------------------------------------------------------------- tier 2e (round 6)
def tier2e():
    """State-independent std head (contextual_std=False) with set_std, and the entropy projections + schedules of
    base_projection_layer.py:14-68 / projection_utils.py:252-280 (reference code under the name-only stubs of tier 2)."""
    from geometry_rl.algorithms.trust_region_projections.projections.base_projection_layer import (
        BaseProjectionLayer, entropy_equality_projection, entropy_inequality_projection)
    from geometry_rl.algorithms.trust_region_projections.utils.projection_utils import get_entropy_schedule
    from geometry_rl.algorithms.trust_region_projections.models.policy.gnn_gaussian_policy_diag import GNNGaussianPolicyDiag

    g = torch.Generator().manual_seed(17)
    B, A = 7, 6

    class FakeGNN(nn.Module):
        device = "cpu"

        def one_step(self, data, input_vector):
            return data

    class FakeData:
        def build_data(self, *args, train=True):
            return self.payload, None

    torch.manual_seed(9)
    fd = FakeData()
    policy = GNNGaussianPolicyDiag(gnn=FakeGNN(), hyper_data=fd, action_dim=A, num_actuators=1, init="orthogonal",
                                   hidden_sizes=(64, 64), contextual_std=False, init_std=0.7, minimal_std=1e-5,
                                   share_action_dim=True, post_fc=False)
    hidden = torch.randn(B, 64, generator=g)
    mean_in = torch.randn(B * 2, 3, generator=g)
    fd.payload = (mean_in, hidden)
    rec = {"hidden": hidden, "gnn_out": mean_in, "pre_std": policy._pre_std.detach().clone()}
    loc, cov = policy(torch.zeros(B, 1), train=True)
    w = torch.rand(B, A, generator=g)
    (cov.diagonal(dim1=-2, dim2=-1) * w).sum().backward()
    rec.update({"loc": loc, "cov": cov, "w": w, "grad.pre_std": policy._pre_std.grad.clone()})
    new_std = (torch.rand(A, generator=g) + 0.2).diag_embed()
    new_std[0, 0] = 0.0   # below the minimal std: clamped (gnn_gaussian_policy_diag.py:140-143)
    policy.set_std(new_std)
    loc2, cov2 = policy(torch.zeros(B, 1), train=True)
    rec.update({"set_std.arg": new_std, "set_std.pre_std": policy._pre_std.detach().clone(), "set_std.cov": cov2})

    # entropy projections on p = (mean, "std" = what the policy returns as covariance)
    mean = torch.randn(B, A, generator=g)
    S = (torch.rand(B, A, generator=g) * 0.6 + 0.05).diag_embed()
    ent = policy.entropy((mean, S))
    beta = ent.mean() + torch.linspace(-1.0, 1.0, B)     # some samples below their bound, some above
    S_g = S.clone().requires_grad_(True)
    pm, pS = entropy_inequality_projection(policy, (mean, S_g), beta)
    wS = torch.rand(B, A, generator=g)
    (pS.diagonal(dim1=-2, dim2=-1) * wS).sum().backward()
    rec.update({"ent.mean": mean, "ent.S": S, "ent.entropy": ent, "ent.beta": beta, "ent.ineq_S": pS, "ent.wS": wS, "ent.ineq_grad_S": S_g.grad.clone()})
    S_g2 = S.clone().requires_grad_(True)
    _, pS2 = entropy_equality_projection(policy, (mean, S_g2), beta)
    (pS2.diagonal(dim1=-2, dim2=-1) * wS).sum().backward()
    rec.update({"ent.eq_S": pS2, "ent.eq_grad_S": S_g2.grad.clone()})
    _, pS3 = entropy_inequality_projection(policy, (mean, S), ent - 1.0)   # nothing to project: returned unchanged
    rec["ent.ineq_noop_S"] = pS3

    # schedules
    steps = torch.tensor([0, 1, 10, 50, 100])
    init_e, target, temp, total = torch.tensor(3.5), torch.tensor(-1.25), 0.5, 100
    for kind in ("linear", "exp"):
        f = get_entropy_schedule(kind, total, dim=A)
        rec[f"sched.{kind}"] = torch.stack([torch.as_tensor(f(init_e, target, temp, int(s_)), dtype=torch.float32) for s_ in steps])
    rec.update({"sched.steps": steps, "sched.initial": init_e, "sched.target": target, "sched.temperature": torch.tensor(temp),
                "sched.total": torch.tensor(total)})
    # the layer's own call with the base (identity) trust-region hook: entropy projection at the scheduled bound, both orders
    for first in (False, True):
        layer = BaseProjectionLayer(proj_type="kl", mean_bound=0.05, cov_bound=0.0025, trust_region_coeff=4.0, scale_prec=True,
                                    entropy_schedule="linear", action_dim=A, total_train_steps=total, target_entropy=float(target),
                                    temperature=temp, entropy_first=first, cpu=True, dtype=torch.float32)
        q = (mean + 0.1, (S.diagonal(dim1=-2, dim2=-1) * 1.3).diag_embed())
        out_m, out_S = layer(policy, (mean, S), q, 40)
        rec[f"layer.first{int(first)}.S"] = out_S
        rec[f"layer.first{int(first)}.initial_entropy"] = layer.initial_entropy
        rec[f"layer.first{int(first)}.bound40"] = torch.as_tensor(layer.get_entropy_bound(40))
    rec["layer.q_S"] = q[1]
    np.savez(os.path.join(OUT, "tier2e_std_entropy.npz"), **npd(rec))
    print("tier2e:", len(rec), "arrays")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "tier2e":
        install_stubs()
        tier2e()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tier2d":   # only the newest tier (the older fixtures stay byte-identical)
        install_stubs()
        tier2d()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "attention":
        install_stubs()
        tier2b(attention=True)
        sys.exit(0)
    tier1()
    install_stubs()
    tier2()
    tier2b()
    tier2c()
    tier2d()
    tier2b(attention=True)
    tier2e()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
