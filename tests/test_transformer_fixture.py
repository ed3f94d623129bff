"""BASELINE config 1's actor pinned to the reference on the CPU (no GPU needed: the transformer baseline is stock torch in the reference
and here): tests/golden/tier2d_transformer_post_fc.npz holds the reference TransformerVanilla + GNNGaussianPolicyDiag(post_fc=True)
state_dict, an input, the (loc, covariance) output and every parameter gradient (tools/make_golden.py tier2d).  The package's modules
must load that state_dict STRICTLY by name and reproduce outputs and gradients."""
import os

import numpy as np
import torch


def test_transformer_actor_matches_reference_fixture(golden_dir):
    from geometry_rl_amd.policy import GNNGaussianPolicyDiag
    from geometry_rl_amd.transformer import TransformerVanilla
    z = np.load(os.path.join(golden_dir, "tier2d_transformer_post_fc.npz"))
    B, P, G = int(z["B"]), int(z["P"]), int(z["G"])
    u_obj, u_grip = torch.from_numpy(z["u_object_geometry"]), torch.from_numpy(z["u_grippers"])
    d = u_obj.shape[1]

    class Graph:
        batch_size = B
        node_types = ["object_geometry", "grippers"]
        nodes_per_sample = {"object_geometry": P, "grippers": G}
        output_mask_key = "grippers"

    class FakeData:   # HyperData(concat_input_vector=True) hands over the dense [B, n, d] tensor
        def build_data(self, *args, train=True):
            return Graph(), torch.cat([u_obj.reshape(B, P, d), u_grip.reshape(B, G, d)], dim=1)

    gnn = TransformerVanilla(input_dim_node=d, output_dim=64, num_layers=2, num_heads=2, hidden_dim=64, dropout=0.0, device="cpu")
    policy = GNNGaussianPolicyDiag(gnn=gnn, hyper_data=FakeData(), action_dim=6, num_actuators=G, contextual_std=True, post_fc=True)
    sd = {k[len("param."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")}
    missing, unexpected = policy.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    loc, cov = policy(torch.zeros(B, 1), train=True)
    assert (loc - torch.from_numpy(z["loc"])).abs().max() <= 1e-5
    assert (cov - torch.from_numpy(z["cov"])).abs().max() <= 1e-5 * float(z["cov"].max())
    (loc * torch.from_numpy(z["w_loc"])).sum().add((cov.diagonal(dim1=-2, dim2=-1) * torch.from_numpy(z["w_cov"])).sum()).backward()
    n = 0
    for k, p in policy.named_parameters():
        key = "grad." + k
        if key in z.files:
            ref = torch.from_numpy(z[key])
            assert (p.grad - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max())), k
            n += 1
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    assert n >= 30
    # the per-type dict form of the reference's construct_input_vector (rigid_tasks_data.py:227-228) is accepted too
    out = gnn.one_step(Graph(), {"object_geometry": u_obj, "grippers": u_grip})
    assert out.shape == (B * G, 64)
