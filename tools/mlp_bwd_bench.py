"""Kernel-only timing of the node-MLP backward (grl_node_mlp_bwd) at the bench's shape, for A/B of builds on one box:
   GRL_LIB=_variants/lib_x.so python tools/mlp_bwd_bench.py [n_nodes]
With a -DGRL_M16_PHASE build also prints the s_memtime shares of the stages of an iteration (wave 0 of every workgroup)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import hip
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536          # rigid HEPi, 4096 frames: 65 536 object nodes in round 1
g = torch.Generator().manual_seed(0)
x2, dout = (torch.randn(n, 16, 64, generator=g).to(dev) for _ in range(2))
w3, b3, w4, b4 = (torch.randn(s, generator=g).mul(0.1).to(dev) for s in [(256, 64), (256,), (64, 256), (64,)])
gam, bet = torch.ones(64, device=dev), torch.zeros(64, device=dev)
rows = n * 16
blocks = hip.query("grl_node_mlp_bwd_blocks", rows)
partial = torch.empty(blocks + 1, hip.query("grl_node_mlp_partial_size"), device=dev)
dx2 = torch.empty_like(x2)
run = lambda: hip.call("grl_node_mlp_bwd", x2, dout, w3, b3, w4, b4, gam, bet, dx2, partial, rows)
for _ in range(3):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
flop = 10 * 64 * 256 * rows
print(f"{os.path.basename(os.environ.get('GRL_LIB', 'libgrl_hip.so')):24s} n_nodes {n}: median {1e3 * ts[5]:8.1f} us  min {1e3 * ts[0]:8.1f} us  "
      f"{flop / ts[5] / 1e9:6.1f} TFLOP/s  checksum {float(dx2.double().abs().sum()):.6e} {float(partial[:blocks].double().sum()):.6e}")
if hasattr(hip.lib(), "grl_mlp16_phase_read"):
    buf = (ctypes.c_ulonglong * 16)()
    hip.lib().grl_mlp16_phase_read(buf, ctypes.c_int(1))
    run(); torch.cuda.synchronize()
    hip.lib().grl_mlp16_phase_read(buf, ctypes.c_int(1))
    names = ["barrier wait", "fragment reads, prefetch issued", "z: 24 MFMA, three GELU tiles", "dH: 24 MFMA, last GELU tile, dZ",
             "split + staging of dZ, transposed reads requested", "dA: 24 MFMA, split of h, row means (next chunk)",
             "partial dA rows written, h staged, reads requested", "dW3: 12 MFMA 32x32 + stage 4 (previous chunk)",
             "dW4: 12 MFMA 32x32 + stage 1 (next chunk)"]
    v = [buf[i] for i in range(9)]
    tot = sum(v) or 1
    per_it = tot / (blocks * ((rows // 16 + blocks - 1) // blocks))
    print(f"   s_memtime ticks per iteration (wave 0): {per_it:.0f}")
    for nme, x in zip(names, v):
        print(f"   {nme:56s} {100 * x / tot:5.1f} %  {x / tot * per_it:7.0f}")
