"""Micro-benchmark of grl_reduce_partials_multi on slabs shaped like the step's (GB/s of slab bytes read)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geometry_rl_amd import hip
dev = torch.device("cuda:0")

def run(name, slabs, segs):
    """slabs: list of (rows, ld); segs: list of (slab index, start, len, dst key)"""
    P = [torch.randn(r, ld, device=dev) for r, ld in slabs]
    dsts = {}
    for _, _, ln, key in segs:
        dsts.setdefault(key, torch.zeros(ln, device=dev))
    n = len(segs)
    args = ((ctypes.c_void_p * n)(*[P[i].data_ptr() for i, _, _, _ in segs]), (ctypes.c_int * n)(*[slabs[i][0] for i, _, _, _ in segs]),
            (ctypes.c_int * n)(*[slabs[i][1] for i, _, _, _ in segs]), (ctypes.c_int * n)(*[s for _, s, _, _ in segs]),
            (ctypes.c_int * n)(*[l for _, _, l, _ in segs]), (ctypes.c_void_p * n)(*[dsts[k].data_ptr() for _, _, _, k in segs]))
    for _ in range(3):
        hip.call("grl_reduce_partials_multi", n, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.call("grl_reduce_partials_multi", n, *args)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    byts = sum(slabs[i][0] * l * 4 for i, _, l, _ in segs)
    print(f"{name:40s} {ms*1e3:8.1f} us  {byts/1e6:8.1f} MB  {byts/ms/1e6:8.1f} GB/s")

E = 9344
edge = [(0, 896, "w1"), (896, 64, "b1"), (960, 4096, "w2"), (5056, 64, "b2")]
run("edge slab, Wk only (1024 rows)", [(1024, E)], [(0, 5120, 4096, "wk")])
run("edge slab, all 5 segments", [(1024, E)], [(0, s, l, k) for s, l, k in edge] + [(0, 5120, 4096, "wk")])
run("two edge slabs, shared basis MLP", [(1024, E), (1024, E)],
    [(i, s, l, k) for i in (0, 1) for s, l, k in edge] + [(0, 5120, 4096, "wk0"), (1, 5120, 4096, "wk1")])
M = 33216
run("mlp slab (256 rows)", [(256, M)], [(0, 0, 16384, "w3"), (0, 16384, 256, "b3"), (0, 16640, 16384, "w4"), (0, 33024, 64, "b4"),
                                         (0, 33088, 64, "g"), (0, 33152, 64, "be")])
run("mlp slab W3 only", [(256, M)], [(0, 0, 16384, "w3")])
run("tall thin: 4096 rows x 4096", [(4096, 4096)], [(0, 0, 4096, "a")])
run("square-ish 1024 x 16384", [(1024, 16384)], [(0, 0, 16384, "a")])

# ---- the step's whole folding launch at a 512-frame shard (slab shapes of the deferred list, tools: ops.flush_deferred_grads)
mlp = [(0, 16384, "w3"), (16384, 256, "b3"), (16640, 16384, "w4"), (33024, 64, "b4"), (33088, 64, "g"), (33152, 64, "be")]
E2 = 9216
slabs = [(256, M), (256, E2), (256, M), (256, E2), (128, 4356), (128, 8448), (128, 1024), (1024, 448), (128, 448), (64, 12608)]
segs = []
for si, tag in ((0, "a"), (2, "b")):
    segs += [(si, s_, l, k + tag) for s_, l, k in mlp]
for si, tag in ((1, "a"), (3, "b")):
    segs += [(si, s_, l, k) for s_, l, k in edge] + [(si, 5120, 4096, "wk" + tag)]
segs += [(4, 0, 4096, "dsw4"), (5, 0, 4096, "dsw3"), (5, 4160, 4096, "dsw2"), (6, 0, 960, "dsw1"), (7, 0, 448, "enc"), (8, 0, 448, "enc"),
         (9, 0, 4096, "fb0"), (9, 4096, 4096, "fb1"), (9, 8192, 4096, "fb2")]
run("whole fold, 512-frame shard", slabs, segs)
run("  only the two MLP slabs", slabs, [s_ for s_ in segs if s_[0] in (0, 2)])
run("  only the two edge slabs", slabs, [s_ for s_ in segs if s_[0] in (1, 3)])
run("  the rest", slabs, [s_ for s_ in segs if s_[0] >= 4])

# ---- round 5: would a COLUMN-BLOCKED slab ([64-column block][row][64]: a block's rows contiguous, 256 B each) fold faster than the
#      row-major one (rows 133 KB apart)?  A blocked slab's block is a row-major [rows][64] matrix of its own, so the existing kernel can
#      read 64 of them as 64 segments: the same 4096 columns x 256 rows both ways.
run("row-major: 4096 cols of a 256 x 33216 slab", [(256, M)], [(0, 0, 4096, "a")])
run("blocked:   64 blocks of [256][64]", [(256, 64)] * 64, [(i, 0, 64, f"k{i}") for i in range(64)])
run("row-major: 4096 cols of a 1024 x 33216 slab", [(1024, M)], [(0, 0, 4096, "a")])
run("blocked:   64 blocks of [1024][64]", [(1024, 64)] * 64, [(i, 0, 64, f"k{i}") for i in range(64)])
