"""Kernel-only timing of the HBM-bound node-side kernels (fiber convolution fwd / bwd, lift + encode fwd / bwd, slab reductions) at the
bench's shape, with the GB/s of their algorithmic traffic -- for A/B of builds on one box:
   GRL_LIB=_variants/lib_x.so python tools/stream_bench.py [n_nodes]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import hip
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 43008          # rigid HEPi, 4096 frames: ~43 k nodes (object points + actuators)
g = torch.Generator().manual_seed(0)
x1, d2 = (torch.randn(n, 16, 64, generator=g).to(dev) for _ in range(2))
fk = torch.randn(16, 16, 64, generator=g).to(dev)
bias = torch.randn(64, generator=g).to(dev)
x2, dx1 = torch.empty_like(x1), torch.empty_like(x1)
S, V = 4, 3
scal, vec = torch.randn(n, S, generator=g).to(dev), torch.randn(n, V, 3, generator=g).to(dev)
grid3 = torch.nn.functional.normalize(torch.randn(16, 3, generator=g), dim=-1).to(dev)
wenc = torch.randn(64, S + V, generator=g).to(dev)
fb = hip.query("grl_fiber_bwd_blocks", n)
fpart = torch.empty(fb, hip.query("grl_fiber_partial_size"), device=dev)
lb = hip.query("grl_lift_bwd_blocks", n)
lpart = torch.empty(lb, 64 * (S + V), device=dev)
NB = n * 16 * 64 * 4
cases = {
    "fiber_conv_fwd": (lambda: hip.call("grl_fiber_conv_fwd", x1, fk, bias, x2, n), 2 * NB, lambda: x2),
    "fiber_conv_bwd": (lambda: hip.call("grl_fiber_conv_bwd", x1, fk, d2, dx1, fpart, n), 3 * NB, lambda: (dx1, fpart)),
    "lift_encode_fwd": (lambda: hip.call("grl_lift_encode_fwd", scal, vec, grid3, wenc, x2, n, S, V), NB, lambda: x2),
    "lift_encode_bwd": (lambda: hip.call("grl_lift_encode_bwd", scal, vec, grid3, d2, lpart, n, S, V), NB, lambda: lpart),
}
if os.environ.get("GRL_BF16"):   # the bf16-storage twins (config 5): latents stored as bf16, half the bytes
    xb, db = x1.bfloat16(), d2.bfloat16()
    x2b, dx1b = torch.empty_like(xb), torch.empty_like(xb)
    NBh = NB // 2
    cases = {
        "fiber_conv_fwd_bf16": (lambda: hip.call("grl_fiber_conv_fwd_bf16", xb, fk, bias, x2b, n), 2 * NBh, lambda: x2b.float()),
        "fiber_conv_bwd_bf16": (lambda: hip.call("grl_fiber_conv_bwd_bf16", xb, fk, db, dx1b, fpart, n), 3 * NBh, lambda: (dx1b.float(), fpart)),
        "lift_encode_fwd_bf16": (lambda: hip.call("grl_lift_encode_fwd_bf16", scal, vec, grid3, wenc, x2b, n, S, V), NBh, lambda: x2b.float()),
        "lift_encode_bwd_bf16": (lambda: hip.call("grl_lift_encode_bwd_bf16", scal, vec, grid3, db, lpart, n, S, V), NBh, lambda: lpart),
    }
only = os.environ.get("GRL_ONLY")
for name, (run, nbytes, outs) in cases.items():
    if only and only not in name:
        continue
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    o = outs()
    o = o if isinstance(o, tuple) else (o,)
    cs = " ".join(f"{float(t.double().sum()):.9e}" for t in o)
    print(f"{os.path.basename(os.environ.get('GRL_LIB', 'libgrl_hip.so')):22s} {name:16s} n {n}: median {1e3 * ts[7]:7.1f} us  min {1e3 * ts[0]:7.1f} us  "
          f"{nbytes / ts[7] / 1e9:5.2f} TB/s  checksum {cs}")
