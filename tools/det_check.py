import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import ops, hepi
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (ns, nd, E) in [(300, 300, 900), (50, 9, 211), (64, 32, 64), (4096*16, 4096*16, 4096*48)]:
    ei = torch.stack([torch.randint(0, ns, (E,), generator=g), torch.randint(0, nd, (E,), generator=g)])
    es = ops.build_edge_set(ei.to(d), ns, nd)
    x = torch.randn(ns, 16, 64, generator=g).to(d)
    ps, pd = torch.rand(ns, 3, generator=g).to(d), torch.rand(nd, 3, generator=g).to(d)
    grid3 = hepi.make_grid(3, 16).to(d).contiguous()
    w1, b1, w2, b2, wk = [t.to(d) for t in (torch.randn(64, 14, generator=g) / 4, torch.randn(64, generator=g), torch.randn(64, 64, generator=g) / 8,
                                             torch.randn(64, generator=g), torch.randn(64, 64, generator=g) / 8)]
    outs = [ops.EdgeConv.apply(x, ps, pd, grid3, w1, b1, w2, b2, wk, es, 3) for _ in range(3)]
    print("edge fwd", (ns, nd, E), [(outs[0] - o).abs().max().item() for o in outs[1:]])
    x2 = torch.randn(nd, 16, 64, generator=g).to(d); xd = torch.randn(nd, 16, 64, generator=g).to(d)
    gam, bet = torch.ones(64, device=d), torch.zeros(64, device=d)
    w3, b3, w4, b4 = [t.to(d) for t in (torch.randn(256, 64, generator=g) / 8, torch.randn(256, generator=g), torch.randn(64, 256, generator=g) / 16, torch.randn(64, generator=g))]
    o2 = [ops.NodeMLP.apply(x2, xd, gam, bet, w3, b3, w4, b4, None) for _ in range(3)]
    print("mlp fwd", nd, [(o2[0] - o).abs().max().item() for o in o2[1:]])
    o3 = [ops.NodeMLP.apply(x2, xd, gam, bet, w3, b3, w4, b4, xd) for _ in range(3)]
    print("mlp fwd acc", nd, [(o3[0] - o).abs().max().item() for o in o3[1:]])
