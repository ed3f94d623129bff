"""Kernel-only timing of the six DeepSets-critic stages (ops.DeepSetsPipeline) at a bench workload's shape, for A/B of builds on one box:
   GRL_LIB=_variants/lib_x.so python tools/critic_bench.py [B n d]        (rigid HEPi: 4096 33 15;  cloth: 4096 239 13)
Prints the median of each stage, their sum, and checksums of the value / every partial slab."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import hip, ops
dev = torch.device("cuda:0")
B, n, d = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 33, 15)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, n, d, generator=g).to(dev)
shapes = [(64, d), (64,), (64,), (64,), (64, 64), (64,), (64, 64), (64,), (64,), (64,), (64, 64), (64,), (1, 64), (1,)]
params = [(torch.randn(s, generator=g) * (0.3 if len(s) > 1 else 0.1)).to(dev) for s in shapes]
params[2] = params[2] + 1.0; params[8] = params[8] + 1.0
dvalue = torch.randn(B, generator=g).to(dev)
pipe = ops.DeepSetsPipeline(x, params, 1)
leaves = [p.clone().requires_grad_(True) for p in params]
stages = [("fwd1", pipe.fwd1), ("fwd2", pipe.fwd2), ("fwd3", pipe.fwd3), ("bwd3", lambda: pipe.bwd3(dvalue)), ("bwd2", pipe.bwd2),
          ("bwd1", lambda: hip.call("grl_deepsets_bwd1", pipe.x, pipe.h1, pipe.stats1, ops.ctypes_double(pipe.c1), pipe.q1, pipe.bst1, pipe.part1,
                                    pipe.B, pipe.n, pipe.d))]
for _ in range(2):
    for _, f in stages:
        f()
torch.cuda.synchronize()
tot = 0.0
line = []
for name, f in stages:
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); tot += ts[7]
    line.append(f"{name} {ts[7]:6.1f}")
cs = [float(pipe.value.double().sum()), float(pipe.part3.double().sum()), float(pipe.part2.double().sum()), float(pipe.part1.double().sum()),
      float(pipe.q1.double().abs().sum())]
print(f"{os.path.basename(os.environ.get('GRL_LIB', 'libgrl_hip.so')):16s} B {B} n {n} d {d}: " + "  ".join(line) + f"  | sum {tot:7.1f} us | " + " ".join(f"{c:.8e}" for c in cs))
