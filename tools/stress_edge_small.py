"""Run-to-run determinism of the edge convolution at the unit-test shapes (small graphs, many source nodes without edges): every
repetition must be bitwise identical to the first.  GRL_REPS repetitions per shape; exit code 1 on any difference."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import ops, hepi
d = torch.device("cuda:0")
REPS = int(os.environ.get("GRL_REPS", "200"))
bad_total = 0
for (ns, nd, E) in [(1500, 1500, 2500), (300, 300, 900), (5000, 5000, 9000), (37, 37, 150)]:
    g = torch.Generator().manual_seed(E)
    src = torch.randint(0, ns, (E,), generator=g); dst = torch.randint(0, nd, (E,), generator=g)
    dst[: min(nd, E)] = torch.arange(min(nd, E))
    ei = torch.stack([src, dst])
    es = ops.build_edge_set(ei.to(d), ns, nd)
    x = torch.randn(ns, 16, 64, generator=g).to(d)
    ps, pd = (torch.rand(ns, 3, generator=g) * 2 - 1).to(d), (torch.rand(nd, 3, generator=g) * 2 - 1).to(d)
    grid3 = hepi.make_grid(3, 16, True).to(d).contiguous()
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(d)
    ew = [rnd(64, 14, sc=0.25), rnd(64), rnd(64, 64, sc=0.125), rnd(64), rnd(64, 64, sc=0.125)]
    dy = torch.randn(nd, 16, 64, generator=g).to(d)

    def edge():
        xs = x.clone().requires_grad_(True)
        ws = [w.clone().requires_grad_(True) for w in ew]
        y = ops.EdgeConv.apply(xs, ps, pd, grid3, *ws, es, 3)
        y.backward(dy)
        return [y.detach(), xs.grad] + [w.grad for w in ws]

    labels = ["x1", "dx_src", "dW1", "db1", "dW2", "db2", "dWk"]
    ref = edge(); torch.cuda.synchronize()
    nb = {}
    for i in range(1, REPS):
        cur = edge(); torch.cuda.synchronize()
        for lab, a, b in zip(labels, ref, cur):
            n = int((a != b).sum())
            if n:
                nb[lab] = nb.get(lab, 0) + 1
                if nb[lab] <= 3:
                    idx = (a != b).nonzero()[:4].tolist()
                    print(f"shape {(ns, nd, E)} rep {i}: {lab} differs in {n} elements, max {float((a - b).abs().max()):.3e}, first {idx}")
    print("shape", (ns, nd, E), "repetitions with a difference per tensor:", nb)
    bad_total += sum(nb.values())
print("n bad total", bad_total)
sys.exit(1 if bad_total else 0)
