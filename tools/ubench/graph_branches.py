"""How does this HIP runtime replay a captured graph with two branches?  Host time inside replay() and wall time per replay for
  A  one stream, 40 small kernels
  B  two independent chains (20 + 20 kernels), forked at the start, joined at the end only
  C  as B plus a join + fork in the middle (main waits for side, side waits for main)
  D  as B, but the side chain is long (one big kernel) and the main chain short: does the end join cost host time?
(torch elementwise kernels of ~4 us each.)"""
import time, torch
dev = torch.device("cuda:0")
x = torch.zeros(4096, device=dev); y = torch.zeros(4096, device=dev); big = torch.zeros(64 << 20, device=dev)
side = torch.cuda.Stream()

def build(kind):
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=cap):
        cur = torch.cuda.current_stream()
        if kind == "A":
            for i in range(40): x.add_(1.0)
        else:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                if kind == "D":
                    for i in range(10): big.add_(1.0)
                else:
                    for i in range(10): y.add_(1.0)
            for i in range(10): x.add_(1.0)
            if kind == "C":
                cur.wait_stream(side); side.wait_stream(cur)
            with torch.cuda.stream(side):
                for i in range(10): y.add_(1.0)
            for i in range(10): x.add_(1.0)
            cur.wait_stream(side)
    return g

for kind in "ABCD":
    g = build(kind)
    for i in range(5): g.replay()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n): g.replay()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    ta = time.perf_counter() - t0
    print(f"{kind}: wall {1e6 * ta / n:7.1f} us per replay, host {1e6 * th / n:7.1f} us")
