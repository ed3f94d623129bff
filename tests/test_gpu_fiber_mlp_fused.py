"""The fiber convolution as the prologue of the ConvNeXt forward (csrc/node_mlp.hip fiber_rot, include/grl_hip.h grl_fiber_node_mlp_fwd;
reference conv.py:88-90,108-109 + 64-69,112): x2[n,p,c] = bias[c] + 1/16 sum_o x1[n,o,c] fk[o,p,c] is computed per 32-row tile by DPP row
rotations against the rotated table that grl_fiber_basis_fwd writes behind fk, so x1 is read once and the separate launch disappears.
Checked here: the table the basis launch writes is the documented permutation of fk (bitwise); the fused block against a plain torch fp32
reference and against the two separate launches (values, x2 as kept for the backward, every gradient), node counts around the tile size
(odd counts leave half a tile), with and without the accumulated previous output, fp32 and bf16 build; a whole recorded update with the
fusion on against the same update with it off."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-5


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def check(name, got, ref, tol=TOL):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert np.isfinite(err) and err <= tol * scale, f"{name}: err {err:.3e} > {tol * scale:.3e}"


def rotated_table(fk):
    """include/grl_hip.h (ABI 206): tab[(((k*8 + t)*2 + h)*16 + p)*4 + j] = fk[(p - k) & 15][p][8t + 4h + j] / 16."""
    k, t, h, p, j = torch.meshgrid(torch.arange(16), torch.arange(8), torch.arange(2), torch.arange(16), torch.arange(4), indexing="ij")
    return (fk[(p - k) & 15, p, 8 * t + 4 * h + j] / 16).reshape(16, 16, 64).contiguous()


def test_basis_launch_writes_the_rotated_table():
    from geometry_rl_amd import agent, graph
    d = dev()
    torch.manual_seed(0)
    spec = graph.rigid_spec()
    actor, _, _, _ = agent.build_agent(spec, agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2), device=d)
    hepi = next(m for m in actor.modules() if type(m).__name__ == "HEPi")
    convs = [c for rnd in hepi.processor for _, c in rnd.items()]
    from geometry_rl_amd import ops
    with torch.no_grad():
        fks = ops.fiber_kernels(hepi.fiber_poly(), hepi.fiber_basis_fn, convs)
    torch.cuda.synchronize()
    assert len(fks) == len(convs) > 0
    for c in convs:
        fk = fks[id(c)]
        assert torch.equal(fk.ftab.cpu(), rotated_table(fk.cpu()))


def _inputs(n, seed):
    g = torch.Generator().manual_seed(seed)
    x1 = torch.randn(n, 16, 64, generator=g)
    fk = torch.randn(16, 16, 64, generator=g)
    bias = torch.randn(64, generator=g)
    xd = torch.randn(n, 16, 64, generator=g)
    prev = torch.randn(n, 16, 64, generator=g)
    gam, bet = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    w3 = torch.randn(256, 64, generator=g) / 8
    b3 = torch.randn(256, generator=g) * 0.1
    w4 = torch.randn(64, 256, generator=g) / 16
    b4 = torch.randn(64, generator=g) * 0.1
    R = torch.randn(n, 16, 64, generator=g)
    return [x1, fk, bias, xd, gam, bet, w3, b3, w4, b4, prev], R


NAMES = ["dx1", "dfk", "dbias", "dx_dst", "dgamma", "dbeta", "dW3", "db3", "dW4", "db4", "dprev"]


def _device_block(ts, R, use_prev, fused, prec=""):
    from geometry_rl_amd import ops
    d = dev()
    lat = torch.bfloat16 if prec else torch.float32
    dl = [t.clone().to(d, dtype=lat if i in (0, 3, 10) else torch.float32).requires_grad_(True) for i, t in enumerate(ts)]
    x1, fk, bias, xd, gam, bet, w3, b3, w4, b4, prev = dl
    fkd = fk.detach()
    fk.ftab = rotated_table(fkd.cpu()).to(d)
    old = ops.FUSE_FIBER_MLP
    ops.FUSE_FIBER_MLP = fused
    try:
        out = ops.conv_block(x1, fk, bias, xd, gam, bet, w3, b3, w4, b4, prev if use_prev else None, None, prec, None)
        assert (type(out.grad_fn).__name__ == "FiberNodeMLPBackward") == fused
        x2 = out.grad_fn.saved_tensors[2].clone() if fused else None
        (out.float() * R.to(d)).sum().backward()
    finally:
        ops.FUSE_FIBER_MLP = old
    torch.cuda.synchronize()
    return out.detach(), x2, [t.grad for t in dl]


@pytest.mark.parametrize("n", [1, 2, 3, 33, 9001])
def test_fused_block_fp32(n):
    ts, R = _inputs(n, 7 + n)
    for use_prev in (False, True):
        leaves = [t.clone().requires_grad_(True) for t in ts]
        X1, FK, B, XD, G, Bt, W3, B3, W4, B4, PV = leaves
        x2_ref = torch.einsum("boc,opc->bpc", X1, FK) / 16 + B
        ref = XD + F.linear(F.gelu(F.linear(F.layer_norm(x2_ref, (64,), G, Bt, 1e-5), W3, B3)), W4, B4)
        if use_prev:
            ref = ref + PV
        (ref * R).sum().backward()
        out_f, x2_f, g_f = _device_block(ts, R, use_prev, fused=True)
        out_s, _, g_s = _device_block(ts, R, use_prev, fused=False)
        check("out vs torch", out_f, ref, 1e-4)   # north_star: outputs within 1e-4
        check("x2 kept for the backward", x2_f, x2_ref)
        check("out vs separate launches", out_f, out_s, 1e-5)
        for name, a, b, c in zip(NAMES, g_f, g_s, leaves):
            if name == "dprev" and not use_prev:
                continue
            check(name + " vs torch", a, c.grad, 2e-4)
            check(name + " vs separate launches", a, b, 2e-5)


@pytest.mark.parametrize("n", [1, 3, 33, 9001])
def test_fused_block_bf16_build(n):
    """bf16 storage: the fused forward rounds x2 to bf16 before the LayerNorm (the backward re-reads it as stored), as the separate launches
    do through HBM; the two forms differ by the summation order of the product, i.e. by single roundings of x2 that flip (2^-8 relative on
    those entries), and by what that does downstream."""
    ts, R = _inputs(n, 70 + n)
    r16 = lambda t: t.to(torch.bfloat16).float()
    for i in (0, 3, 10):
        ts[i] = r16(ts[i])
    R = r16(R)
    out_f, x2_f, g_f = _device_block(ts, R, True, fused=True, prec="_bf16")
    out_s, _, g_s = _device_block(ts, R, True, fused=False, prec="_bf16")
    assert out_f.dtype == torch.bfloat16 and x2_f.dtype == torch.bfloat16
    x2_ref = torch.einsum("boc,opc->bpc", ts[0], ts[1]) / 16 + ts[2]
    err = (x2_f.float().cpu() - x2_ref).abs()
    tol = 2.0 ** -8 * x2_ref.abs() + 1e-5 * float(x2_ref.abs().max())
    assert bool((err <= tol).all()), float((err - tol).max())
    # values: bf16 storage of out (2^-8) on top of the product's own noise
    scale = float(out_s.float().abs().max())
    assert float((out_f.float() - out_s.float()).abs().max()) <= 2e-2 * scale
    for name, a, b in zip(NAMES, g_f, g_s):
        a, b = a.float(), b.float()
        assert float((a - b).abs().max()) <= 2e-2 * max(float(b.abs().max()), 1e-6), name


def _update(model, precision, fused, steps=3):
    from geometry_rl_amd import agent, graph, ops, synthetic as syn
    d = dev()
    old = ops.FUSE_FIBER_MLP
    ops.FUSE_FIBER_MLP = fused
    try:
        if model == "empn":
            spec = graph.rigid_spec(G=2, angular_velocity=False, object_velocity=False)
            cfg = agent.AgentConfig(model="empn", precision=precision)
            obs = syn.make_rigid_obs(24, G=2, angular_velocity=False, object_velocity=False, seed=8)
        else:
            spec = graph.rigid_spec()
            cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2, precision=precision)
            obs = syn.make_rigid_obs(24, seed=3)
        torch.manual_seed(0)
        actor, critic, proj, loss = agent.build_agent(spec, cfg, device=d)
        A = spec.num_actuators * cfg.output_dim_vec * 3
        batch = dict(obs)
        batch.update(syn.make_ppo_fields(24, A, seed=5))
        batch = {k: v.to(d) for k, v in batch.items()}
        with torch.no_grad():
            actor.forward_diag(*[batch[k] for k in spec.in_features], train=True)    # calibration
        upd = agent.PolicyUpdater(loss, lr=cfg.lr, use_graph=True)
        outs = []
        for _ in range(steps):   # eager, recording, replay
            o = upd.step(batch)
            outs.append({k: float(v) for k, v in o.items() if torch.is_tensor(v) and v.numel() == 1})
        torch.cuda.synchronize()
        assert upd.mode.startswith("graph")
        return upd.flat.clone(), outs, cfg.lr
    finally:
        ops.FUSE_FIBER_MLP = old


@pytest.mark.parametrize("model", ["hepi", "empn"])
def test_recorded_update_with_and_without_the_fusion(model):
    """Same program, one launch fewer per convolution: every reported value of three updates agrees to the summation order of the depthwise
    product, and so do the parameters -- in units of lr: Adam's first steps move an entry by ~lr * sign(gradient), so an entry whose
    gradient is rounding noise may differ by a whole step; the MEAN difference must be a small fraction of one."""
    pa, oa, lr = _update(model, "fp32", False)
    pb, ob, _ = _update(model, "fp32", True)
    diff = (pa - pb).abs()
    assert float(diff.max()) <= 2.5 * lr * 3 and float(diff.mean()) <= 0.01 * lr, (float(diff.max()), float(diff.mean()), lr)
    for step, (a, b) in enumerate(zip(oa, ob)):
        assert set(a) == set(b)
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-5 * max(1.0, abs(a[k])), (step, k, a[k], b[k])
