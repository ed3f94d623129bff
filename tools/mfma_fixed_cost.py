"""Fixed cost of the MFMA kernels: kernel-only durations (in-library wall-clock stamps, hip.kernel_prof_enable(2)) of the edge forward /
backward and the node-MLP forward / backward over graph sizes from a few hundred nodes up -- the intercept of time against size is what
a launch costs before it does any work (prologue: weight images into LDS; epilogue: accumulator drain, partial rows; tail: the slowest wave).
   python tools/mfma_fixed_cost.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import hip, ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
grid3 = torch.nn.functional.normalize(torch.randn(16, 3, generator=g), dim=-1).to(dev)
w1, b1, w2, b2, wk = (torch.randn(s, generator=g).mul(0.1).to(dev).requires_grad_(True) for s in [(64, 14), (64,), (64, 64), (64,), (64, 64)])
w3, b3, w4, b4 = (torch.randn(s, generator=g).mul(0.1).to(dev).requires_grad_(True) for s in [(256, 64), (256,), (64, 256), (64,)])
gam, bet = torch.ones(64, device=dev, requires_grad=True), torch.zeros(64, device=dev, requires_grad=True)
print(f"{'nodes':>8s} {'edges':>8s} | edge fwd  edge bwd  mlp fwd  mlp bwd   (us, median of 7)")
for n in (64, 256, 1024, 4096, 8192, 16384, 43008):
    E = 3 * n
    src = torch.randint(0, n, (E,), generator=g)
    dst = torch.arange(n).repeat_interleave(3)
    es = ops.build_edge_set(torch.stack([src, dst]).to(dev), n, n)
    x = torch.randn(n, 16, 64, generator=g).to(dev).requires_grad_(True)
    pos = torch.randn(n, 3, generator=g).to(dev)
    R = torch.randn(n, 16, 64, generator=g).to(dev)
    def once():
        x1 = ops.EdgeConv.apply(x, pos, pos, grid3, w1, b1, w2, b2, wk, es, 3, None, "")
        out = ops.NodeMLP.apply(x1, x, gam, bet, w3, b3, w4, b4, None, None, "")
        out.backward(R)
    once(); once()
    torch.cuda.synchronize()
    rec = {}
    for _ in range(7):
        hip.kernel_prof_enable(2)
        once()
        torch.cuda.synchronize()
        for k, (c, ms) in hip.kernel_prof_summary().items():
            rec.setdefault(k, []).append(1e3 * ms / c)
        hip.kernel_prof_enable(False)
    med = {k: sorted(v)[len(v) // 2] for k, v in rec.items()}
    print(f"{n:8d} {E:8d} | " + "  ".join(f"{med.get(k, float('nan')):8.1f}" for k in ("edge_conv_fwd_kernel", "edge_bwd16_kernel", "node_mlp_fwd_kernel", "node_mlp_bwd16_kernel")))
