"""Kernel-only timing of the edge convolution forward (grl_edge_conv_fwd -> edge16_kernel<0>) at the bench's round-1 shape (65 536 nodes,
three in-edges each), for A/B of builds and PMC passes:   GRL_LIB=_variants/lib_x.so python tools/edge_fwd_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import hip, ops, hepi
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ei = torch.stack([torch.randint(0, n, (3 * n,), generator=g), torch.arange(n).repeat_interleave(3)])
es = ops.build_edge_set(ei.to(dev), n, n)
x = torch.randn(n, 16, 64, generator=g).to(dev)
ps = torch.rand(n, 3, generator=g).to(dev)
grid = hepi.make_grid(3, 16, True).to(dev).contiguous()
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
ew = [rnd(64, 14, sc=0.25), rnd(64), rnd(64, 64, sc=0.125), rnd(64), rnd(64, 64, sc=0.125)]
run = lambda: ops.EdgeConv.apply(x, ps, ps, grid, *ew, es, 3, None, "")
with torch.no_grad():
    for _ in range(3):
        y = run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"{os.path.basename(os.environ.get('GRL_LIB', 'libgrl_hip.so')):24s} {n} nodes, {3 * n} edges: median {1e3 * ts[5]:8.1f} us  min {1e3 * ts[0]:8.1f} us  checksum {float(y.double().abs().sum()):.6e}")
