"""Pre-split weight images (csrc/grl_wimg.h, grl_weight_images): a launch that copies an image and one that
stages the weights itself must agree BITWISE -- the image bytes are what the kernels' own staging code writes to LDS -- for the edge
convolution (16-row kernels and the few-tile 32-row forward), the ConvNeXt node block, forward and backward, fp32 and bf16 builds; and a
whole policy update with the images switched on lands on the parameters of one without them."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _weights(g, d):
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(d)
    edge = [r(64, 14, sc=0.25), r(64, sc=0.3), r(64, 64, sc=0.125), r(64, sc=0.3), r(64, 64, sc=0.125)]
    mlp = [1 + r(64, sc=0.1), r(64, sc=0.1), r(256, 64, sc=0.125), r(256, sc=0.1), r(64, 256, sc=0.06), r(64, sc=0.1)]   # gamma beta W3 b3 W4 b4
    return edge, mlp


@pytest.mark.parametrize("prec", ["", "_bf16"])
@pytest.mark.parametrize("n,e", [(40, 160), (700, 2100), (3000, 9000)])   # <= 512 forward tiles: 32-row kernel (kind 1); above: 16-row (kind 0)
def test_image_launches_equal_staging_launches_bitwise(n, e, prec):
    from geometry_rl_amd import hip, ops, hepi
    d = dev()
    g = torch.Generator().manual_seed(n)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)])
    es = ops.build_edge_set(ei.to(d), n, n)
    dt = hip.storage_dtype(prec)
    x = torch.randn(n, 16, 64, generator=g).to(d).to(dt)
    xd = torch.randn(n, 16, 64, generator=g).to(d).to(dt)
    dy = torch.randn(n, 16, 64, generator=g).to(d).to(dt)
    pos = (torch.rand(n, 3, generator=g) * 2 - 1).to(d)
    grid3 = hepi.make_grid(3, 16, True).to(d).contiguous()
    edge_w, mlp_w = _weights(g, d)

    def run(use_images):
        ew = [w.clone().requires_grad_(True) for w in edge_w]
        mw = [w.clone().requires_grad_(True) for w in mlp_w]
        img = None
        if use_images:
            (img,) = ops.weight_images([(ew[4], n, tuple(mw))], grid3, tuple(ew[:4]), prec, with_backward=True)
            assert img is not None and img.e16 is not None and img.mlp_f is not None and img.mlp_b is not None
            assert (img.e32 is not None) == ((n + 1) // 2 <= 512)
        xs = x.clone().requires_grad_(True)
        y = ops.EdgeConv.apply(xs, pos, pos, grid3, *ew, es, 3, None, prec, img)
        y.backward(dy)
        x2 = x.clone().requires_grad_(True)
        z = ops.NodeMLP.apply(x2, xd, *mw, None, None, prec, img)
        z.backward(dy)
        torch.cuda.synchronize()
        return [y.detach(), xs.grad, z.detach(), x2.grad] + [w.grad for w in ew + mw]

    a, b = run(False), run(True)
    names = ["x1", "dx_src", "out", "dx2", "dW1", "db1", "dW2", "db2", "dWk", "dgamma", "dbeta", "dW3", "db3", "dW4", "db4"]
    for nm, u, v in zip(names, a, b):
        assert torch.equal(u, v), f"{nm}: image launch differs from staging launch (max |diff| {(u.float() - v.float()).abs().max().item():.3e})"


def test_image_bytes_are_the_layouts_the_header_states():
    from geometry_rl_amd import hip
    assert hip.query("grl_wimg_bytes", 0) == 2 * 64 * 40 * 2 + 4 * 64 * 80 * 2 + 768 + 4 * 64 * 80 * 2     # Edge16Image
    assert hip.query("grl_wimg_bytes", 1) == 2 * 64 * 24 * 2 + 4 * 64 * 72 * 2 + 768                       # ChainW
    assert hip.query("grl_wimg_bytes", 2) == 2 * 256 * 72 * 2 + 2 * 64 * 264 * 2 + (256 + 3 * 64) * 4      # MlpSmemBf
    assert hip.query("grl_wimg_bytes", 3) == 3 * 4 * 4 * 2 * 2 * 64 * 16                                    # Mlp16Image
    assert hip.query("grl_wimg_bytes", 4) == -1


@pytest.mark.parametrize("model", ["hepi", "empn"])
def test_policy_update_with_images_equals_update_without(model, monkeypatch):
    """Three updates (eager, recorded, replayed) of the same agent with and without the images: identical parameters, bit for bit."""
    from geometry_rl_amd import agent, graph, ops, synthetic as syn
    d = dev()
    B = 48
    G = 2 if model == "empn" else 1
    spec = graph.rigid_spec(G=G) if G > 1 else graph.rigid_spec()
    cfg = agent.AgentConfig(model=model) if model == "empn" else agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
    A = spec.num_actuators * cfg.output_dim_vec * 3
    batches = []
    for i in range(3):
        b = dict(syn.make_rigid_obs(B, G=G, seed=10 + i) if G > 1 else syn.make_rigid_obs(B, seed=10 + i))
        b.update(syn.make_ppo_fields(B, A, seed=20 + i))
        batches.append({k: v.to(d) for k, v in b.items()})

    def run(use):
        monkeypatch.setattr(ops, "USE_WEIGHT_IMAGES", use)
        torch.manual_seed(3)
        actor, critic, proj, loss = agent.build_agent(spec, cfg, device=d)
        upd = agent.PolicyUpdater(loss, use_graph=True)
        for b in batches:
            upd.step(b)
        torch.cuda.synchronize()
        return upd.flat.clone(), upd.mode

    (p0, m0), (p1, m1) = run(False), run(True)
    assert m1.startswith("graph")
    assert torch.equal(p0, p1), f"parameters differ: max |diff| {(p0 - p1).abs().max().item():.3e}"
