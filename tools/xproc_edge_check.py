"""Cross-process determinism of the edge convolution backward at a unit-test shape: the first process stores its outputs, later
processes compare bitwise and print where they differ.   python tools/xproc_edge_check.py <file.pt>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import ops, hepi
d = torch.device("cuda:0")
path = sys.argv[1]
ns = nd = 1500; E = 2500
g = torch.Generator().manual_seed(E)
src = torch.randint(0, ns, (E,), generator=g); dst = torch.randint(0, nd, (E,), generator=g)
dst[:nd] = torch.arange(nd)
ei = torch.stack([src, dst])
es = ops.build_edge_set(ei.to(d), ns, nd)
x = torch.randn(ns, 16, 64, generator=g).to(d)
ps, pd = (torch.rand(ns, 3, generator=g) * 2 - 1).to(d), (torch.rand(nd, 3, generator=g) * 2 - 1).to(d)
grid3 = hepi.make_grid(3, 16, True).to(d).contiguous()
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(d)
ew = [rnd(64, 14, sc=0.25), rnd(64), rnd(64, 64, sc=0.125), rnd(64), rnd(64, 64, sc=0.125)]
dy = torch.randn(nd, 16, 64, generator=g).to(d)
xs = x.clone().requires_grad_(True)
ws = [w.clone().requires_grad_(True) for w in ew]
y = ops.EdgeConv.apply(xs, ps, pd, grid3, *ws, es, 3)
y.backward(dy)
out = {"x1": y.detach().cpu(), "dx_src": xs.grad.cpu(), "rowptr_s": es.rowptr_s.cpu(), "src_s": es.src_s.cpu(), "dst_s": es.dst_s.cpu()}
for i, w in enumerate(ws):
    out[f"dw{i}"] = w.grad.cpu()
if not os.path.exists(path):
    torch.save(out, path); print("stored reference")
else:
    ref = torch.load(path)
    for k in out:
        a, b = ref[k], out[k]
        n = int((a != b).sum())
        if n:
            idx = (a != b).nonzero()
            rows = sorted(set(idx[:, 0].tolist()))
            print(f"{k}: {n} elements differ, max {float((a.double() - b.double()).abs().max()):.3e}; nodes {rows[:12]} ({len(rows)} nodes); orientations {sorted(set(idx[:,1].tolist()))[:16] if idx.shape[1] > 1 else ''}; channels {sorted(set(idx[:,2].tolist()))[:8] if idx.shape[1] > 2 else ''}")
            if k == "dx_src":
                rp = out["rowptr_s"]
                for nd_ in rows[:6]:
                    print("   node", nd_, "out-degree", int(rp[nd_ + 1] - rp[nd_]), "edges", int(rp[nd_]), "..", int(rp[nd_ + 1]), " chunk pos", nd_ % 16 if False else "")
    print("compared")
