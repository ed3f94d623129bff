"""Run-to-run determinism of the four MFMA ops (edge conv, node MLP; forward and backward): every repetition must be bitwise
identical to the first.  Exit code 1 on any difference.  GRL_LIB selects a variant build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import ops, hepi
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
REPS = int(os.environ.get("GRL_REPS", "6"))
ns = nd = 65536; E = 196608
ei = torch.stack([torch.randint(0, ns, (E,), generator=g), torch.randint(0, nd, (E,), generator=g)])
es = ops.build_edge_set(ei.to(d), ns, nd)
x = torch.randn(ns, 16, 64, generator=g).to(d)
ps, pd = torch.rand(ns, 3, generator=g).to(d), torch.rand(nd, 3, generator=g).to(d)
grid3 = hepi.make_grid(3, 16).to(d).contiguous()
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(d)
ew = [rnd(64, 14, sc=0.25), rnd(64), rnd(64, 64, sc=0.125), rnd(64), rnd(64, 64, sc=0.125)]
mw = [1 + rnd(64, sc=0.1), rnd(64, sc=0.1), rnd(256, 64, sc=0.125), rnd(256, sc=0.1), rnd(64, 256, sc=0.06), rnd(64, sc=0.1)]
dy = torch.randn(nd, 16, 64, generator=g).to(d)
xd = torch.randn(nd, 16, 64, generator=g).to(d)


from geometry_rl_amd import hip


def edge(prec=""):
    dt = hip.storage_dtype(prec)
    xs = x.to(dt).clone().detach().requires_grad_(True)
    ws = [w.clone().requires_grad_(True) for w in ew]
    y = ops.EdgeConv.apply(xs, ps, pd, grid3, *ws, es, 3, None, prec)
    y.backward(dy.to(dt))
    return [y.detach(), xs.grad] + [w.grad for w in ws]


def mlp(prec=""):
    dt = hip.storage_dtype(prec)
    x2 = x.to(dt).clone().detach().requires_grad_(True)
    ws = [w.clone().requires_grad_(True) for w in mw]
    y = ops.NodeMLP.apply(x2, xd.to(dt), *ws, None, None, prec)
    y.backward(dy.to(dt))
    return [y.detach(), x2.grad] + [w.grad for w in ws]


bad_total = 0
EL, ML = ["x1", "dx_src", "dW1", "db1", "dW2", "db2", "dWk"], ["out", "dx2", "dgamma", "dbeta", "dW3", "db3", "dW4", "db4"]
for name, fn, labels in (("edge_conv", edge, EL), ("node_mlp", mlp, ML), ("edge_conv_bf16", lambda: edge("_bf16"), EL),
                         ("node_mlp_bf16", lambda: mlp("_bf16"), ML)):
    ref = fn()
    torch.cuda.synchronize()
    for i in range(1, REPS):
        cur = fn()
        torch.cuda.synchronize()
        for lab, a, b in zip(labels, ref, cur):
            nbad = int((a != b).sum())
            if nbad:
                bad_total += nbad
                print(f"{name} rep {i}: {lab} differs in {nbad} elements, max {float((a - b).abs().max()):.3e}")
    print(name, "done")
print("n bad total", bad_total)
sys.exit(1 if bad_total else 0)
