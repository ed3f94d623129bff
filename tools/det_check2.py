import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import ops, hepi
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
ns = nd = 65536; E = 196608
ei = torch.stack([torch.randint(0, ns, (E,), generator=g), torch.randint(0, nd, (E,), generator=g)])
es = ops.build_edge_set(ei.to(d), ns, nd)
x = torch.randn(ns, 16, 64, generator=g).to(d)
ps, pd = torch.rand(ns, 3, generator=g).to(d), torch.rand(nd, 3, generator=g).to(d)
grid3 = hepi.make_grid(3, 16).to(d).contiguous()
w1, b1, w2, b2, wk = [t.to(d) for t in (torch.randn(64, 14, generator=g) / 4, torch.randn(64, generator=g), torch.randn(64, 64, generator=g) / 8,
                                         torch.randn(64, generator=g), torch.randn(64, 64, generator=g) / 8)]
outs = []
for i in range(4):
    outs.append(ops.EdgeConv.apply(x, ps, pd, grid3, w1, b1, w2, b2, wk, es, 3))
    torch.cuda.synchronize()
for i in range(1, 4):
    diff = (outs[0] - outs[i]).abs().amax(dim=(1, 2))
    bad = torch.nonzero(diff > 0).reshape(-1)
    print(i, "n bad nodes", bad.numel(), "first", bad[:10].tolist(), "last", bad[-5:].tolist(), "max", diff.max().item())
deg = (es.rowptr_d[1:] - es.rowptr_d[:-1])
if bad.numel():
    print("deg of bad", deg[bad[:20]].tolist())
    print("outs0 at bad0", outs[0][bad[0], 0, :4].tolist(), "outs1", outs[1][bad[0], 0, :4].tolist())
for b_ in bad[:6].tolist():
    dd = (outs[0][b_] - outs[3][b_]).abs()
    rows = torch.nonzero(dd.amax(dim=1) > 0).reshape(-1).tolist()
    cols = torch.nonzero(dd.amax(dim=0) > 0).reshape(-1).tolist()
    print("node", b_, "deg", int(deg[b_]), "ori rows differing", rows, "n cols", len(cols), "cols", cols[:40])
