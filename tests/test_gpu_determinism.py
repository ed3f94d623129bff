"""Run-to-run determinism of the MFMA kernels (edge convolution, node MLP; forward and backward, weight gradients included).

Two waves share a SIMD in the edge forward and in the node-MLP kernels; a mis-ordered operand load there shows up as a handful
of wrong tiles that differ from run to run (DESIGN.md "MFMA operand hazard"), so every repetition must be BITWISE equal."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(shape="half"):
    """``half``: 32 768 nodes / 98 304 random edges (the round-1 shape); ``internal``: the internal convolution of BASELINE's 4096-frame
    rigid minibatch -- 65 536 nodes, 196 608 edges, three in-edges per node like the kNN graph; ``task``: its object -> gripper
    convolution -- 65 536 sources, 4096 destinations with 16 in-edges each = 2048 destination tiles... at a 1024-frame shard 512 tiles,
    i.e. the SPLIT forward (one workgroup per tile, edge_conv_fwd_kernel<true>)."""
    from geometry_rl_amd import ops, hepi
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    if shape == "half":
        ns = nd = 32768
        E = 98304
        ei = torch.stack([torch.randint(0, ns, (E,), generator=g), torch.randint(0, nd, (E,), generator=g)])
    elif shape == "internal":
        ns = nd = 65536
        dst = torch.arange(nd).repeat_interleave(3)
        ei = torch.stack([torch.randint(0, ns, (3 * nd,), generator=g), dst])
    else:   # task, 1024-frame shard: 16384 sources -> 1024 destinations, 16 in-edges each (512 destination tiles: SPLIT path)
        ns, nd = 16384, 1024
        ei = torch.stack([torch.arange(ns), torch.arange(ns) // 16])
    es = ops.build_edge_set(ei.to(d), ns, nd)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(d)
    t = dict(es=es, x=rnd(ns, 16, 64), ps=torch.rand(ns, 3, generator=g).to(d), pd=torch.rand(nd, 3, generator=g).to(d),
             grid=hepi.make_grid(3, 16).to(d).contiguous(), dy=rnd(nd, 16, 64), xd=rnd(nd, 16, 64),
             ew=[rnd(64, 14, sc=0.25), rnd(64), rnd(64, 64, sc=0.125), rnd(64), rnd(64, 64, sc=0.125)],
             mw=[1 + rnd(64, sc=0.1), rnd(64, sc=0.1), rnd(256, 64, sc=0.125), rnd(256, sc=0.1), rnd(64, 256, sc=0.06), rnd(64, sc=0.1)])
    return ops, t


def _edge(ops, t, prec=""):
    from geometry_rl_amd import hip
    dt = hip.storage_dtype(prec)   # the "_bf16" kernels keep the node latents in bf16
    xs = t["x"].to(dt).requires_grad_(True)
    ws = [w.clone().requires_grad_(True) for w in t["ew"]]
    y = ops.EdgeConv.apply(xs, t["ps"], t["pd"], t["grid"], *ws, t["es"], 3, None, prec)
    y.backward(t["dy"].to(dt))
    return [y.detach(), xs.grad] + [w.grad for w in ws]


def _mlp(ops, t, prec=""):
    from geometry_rl_amd import hip
    dt = hip.storage_dtype(prec)
    x2 = t["x"][:t["xd"].shape[0]].to(dt).requires_grad_(True)
    ws = [w.clone().requires_grad_(True) for w in t["mw"]]
    y = ops.NodeMLP.apply(x2, t["xd"].to(dt), *ws, None, None, prec)
    y.backward(t["dy"].to(dt))
    return [y.detach(), x2.grad] + [w.grad for w in ws]


@pytest.mark.parametrize("which,shape,prec", [("edge_conv", "half", ""), ("node_mlp", "half", ""),
                                              ("edge_conv", "internal", ""), ("edge_conv", "task", ""), ("node_mlp", "internal", ""),
                                              ("edge_conv", "internal", "_bf16"), ("node_mlp", "internal", "_bf16")])
def test_bitwise_reproducible(which, shape, prec):
    ops, t = _setup(shape)
    if shape == "task":   # the launcher must have taken the one-workgroup-per-tile forward for this shape
        assert (t["es"].n_dst + 1) // 2 <= 512
    fn_ = _edge if which == "edge_conv" else _mlp
    fn = lambda o, tt: fn_(o, tt, prec)
    ref = fn(ops, t)
    torch.cuda.synchronize()
    for rep in range(8 if shape == "half" else 5):
        cur = fn(ops, t)
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(ref, cur)):
            assert torch.equal(a, b), f"{which}: output {k} differs in repetition {rep} ({int((a != b).sum())} elements)"
