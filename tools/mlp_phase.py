"""Phase timing of node_mlp_bwd_fused_kernel (debug build with -DGRL_MLP_PHASE_PROF, GRL_LIB=<that .so>)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import hip
dev = torch.device("cuda:0")
n = int(os.environ.get("ROWS", 2097152))
g = torch.Generator().manual_seed(0)
mk = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(dev)
x2, dout = mk(n, 64), mk(n, 64)
W3, b3, W4, b4, gam, bet = mk(256, 64), mk(256), mk(64, 256), mk(64), mk(64) + 1, mk(64)
dx2 = torch.empty_like(x2)
blocks = hip.query("grl_node_mlp_bwd_blocks", n)
partial = torch.empty(blocks + 1, hip.query("grl_node_mlp_partial_size"), device=dev)   # last row: scratch
buf = (ctypes.c_ulonglong * 32)()
for it in range(3):
    hip.call("grl_node_mlp_bwd", x2, dout, W3, b3, W4, b4, gam, bet, dx2, partial, n)
torch.cuda.synchronize()
hip.lib().grl_mlp_phase_read(buf, ctypes.c_int(1))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); hip.call("grl_node_mlp_bwd", x2, dout, W3, b3, W4, b4, gam, bet, dx2, partial, n); e1.record(); torch.cuda.synchronize()
hip.lib().grl_mlp_phase_read(buf, ctypes.c_int(1))
print("kernel ms", e0.elapsed_time(e1), "blocks", blocks, "chunks/block", n / 32 / blocks)
names = ["loop top", "P1 LN+images", "barrier A", "P2 transposes", "z,dH MFMA", "gelu+splits", "barrier B", "dW3,dW4", "dA compute", "DA store+barrier C",
         "DA add+barrier D", "P4 LN bwd"]
for w in range(2):
    tot = sum(buf[w * 16 + i] for i in range(12))
    print("wave", 0 if w == 0 else 5, "total ticks/block", tot / blocks)
    for i in range(12):
        print(f"   {names[i]:22s} {buf[w * 16 + i] / blocks / (n / 32 / blocks):9.1f} ticks/chunk  {100 * buf[w * 16 + i] / tot:5.1f} %")
