"""Run-to-run determinism of the MFMA kernels (edge convolution, node MLP; forward and backward, weight gradients included).

Two waves share a SIMD in the edge forward and in the node-MLP kernels; a mis-ordered operand load there shows up as a handful
of wrong tiles that differ from run to run (DESIGN.md "MFMA operand hazard"), so every repetition must be BITWISE equal."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup():
    from geometry_rl_amd import ops, hepi
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    ns = nd = 32768
    E = 98304
    ei = torch.stack([torch.randint(0, ns, (E,), generator=g), torch.randint(0, nd, (E,), generator=g)])
    es = ops.build_edge_set(ei.to(d), ns, nd)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(d)
    t = dict(es=es, x=rnd(ns, 16, 64), ps=torch.rand(ns, 3, generator=g).to(d), pd=torch.rand(nd, 3, generator=g).to(d),
             grid=hepi.make_grid(3, 16).to(d).contiguous(), dy=rnd(nd, 16, 64), xd=rnd(nd, 16, 64),
             ew=[rnd(64, 14, sc=0.25), rnd(64), rnd(64, 64, sc=0.125), rnd(64), rnd(64, 64, sc=0.125)],
             mw=[1 + rnd(64, sc=0.1), rnd(64, sc=0.1), rnd(256, 64, sc=0.125), rnd(256, sc=0.1), rnd(64, 256, sc=0.06), rnd(64, sc=0.1)])
    return ops, t


def _edge(ops, t):
    xs = t["x"].clone().requires_grad_(True)
    ws = [w.clone().requires_grad_(True) for w in t["ew"]]
    y = ops.EdgeConv.apply(xs, t["ps"], t["pd"], t["grid"], *ws, t["es"], 3)
    y.backward(t["dy"])
    return [y.detach(), xs.grad] + [w.grad for w in ws]


def _mlp(ops, t):
    x2 = t["x"].clone().requires_grad_(True)
    ws = [w.clone().requires_grad_(True) for w in t["mw"]]
    y = ops.NodeMLP.apply(x2, t["xd"], *ws, None)
    y.backward(t["dy"])
    return [y.detach(), x2.grad] + [w.grad for w in ws]


@pytest.mark.parametrize("which", ["edge_conv", "node_mlp"])
def test_bitwise_reproducible(which):
    ops, t = _setup()
    fn = _edge if which == "edge_conv" else _mlp
    ref = fn(ops, t)
    torch.cuda.synchronize()
    for rep in range(8):
        cur = fn(ops, t)
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(ref, cur)):
            assert torch.equal(a, b), f"{which}: output {k} differs in repetition {rep} ({int((a != b).sum())} elements)"
