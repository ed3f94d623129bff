"""GPU parity of each HIP op (through the C-ABI library) against the CPU oracle on identical seeded inputs.

Tolerance: fp32 kernels vs fp32 oracle, max-abs error <= 1e-4 * max(1, |ref|_max) (north_star: outputs within 1e-4)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import equivariant as eq

pytestmark = pytest.mark.gpu

TOL = 1e-4


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def check(name, got, ref, tol=TOL):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    print(f"{name}: max|err|={err:.3e} ref_max={ref.abs().max().item():.3e}")
    assert np.isfinite(err) and err <= tol * scale, f"{name}: err {err:.3e} > {tol * scale:.3e}"


def rand_graph(g, n_src, n_dst, n_edges, bipartite):
    src = torch.randint(0, n_src, (n_edges,), generator=g)
    dst = torch.randint(0, n_dst, (n_edges,), generator=g)
    dst[: min(n_dst, n_edges)] = torch.arange(min(n_dst, n_edges))  # every dst gets at least one edge when possible
    return torch.stack([src, dst])


def params(g, shapes):
    return [torch.randn(*s, generator=g) * (1.0 / np.sqrt(s[-1])) for s in shapes]


@pytest.mark.parametrize("n_src,n_dst,n_edges,dim,upper", [(37, 37, 150, 3, False), (50, 9, 211, 3, True), (21, 5, 1, 2, False),
                                                             (300, 300, 900, 3, False), (1500, 1500, 2500, 3, True)])  # last: > 512 tiles = the one-wave-per-tile forward
def test_edge_conv(n_src, n_dst, n_edges, dim, upper):
    from geometry_rl_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(n_edges)
    grid = eq.make_grid(dim, 16, upper)
    grid3 = F.pad(grid, (0, 3 - grid.shape[1]))
    ei = rand_graph(g, n_src, n_dst, n_edges, n_src != n_dst)
    x_src = torch.randn(n_src, 16, 64, generator=g)
    pos_s, pos_d = torch.rand(n_src, 3, generator=g) * 2 - 1, torch.rand(n_dst, 3, generator=g) * 2 - 1
    w1, b1, w2, b2, wk = params(g, [(64, 14), (64,), (64, 64), (64,), (64, 64)])
    R = torch.randn(n_dst, 16, 64, generator=g)

    # oracle
    leaves = [t.clone().requires_grad_(True) for t in (x_src, w1, b1, w2, b2, wk)]
    xs, W1, B1, W2, B2, WK = leaves
    P = {"b.1.weight": W1, "b.1.bias": B1, "b.3.weight": W2, "b.3.bias": B2}
    ps, pd = pos_s[ei[0]], pos_d[ei[1]]
    if dim == 2:
        ps, pd = ps[:, :2], pd[:, :2]
    kb = eq.basis_mlp(eq.spatial_invariants(grid, ps, pd), P, "b")
    x1_ref = eq.scatter_sum(F.linear(kb, WK) * xs[ei[0]], ei[1], n_dst)
    (x1_ref * R).sum().backward()

    es = ops.build_edge_set(ei.to(d), n_src, n_dst)
    dl = [t.clone().to(d).requires_grad_(True) for t in (x_src, w1, b1, w2, b2, wk)]
    x1 = ops.EdgeConv.apply(dl[0], pos_s.to(d), pos_d.to(d), grid3.to(d), dl[1], dl[2], dl[3], dl[4], dl[5], es, dim)
    check("x1", x1, x1_ref)
    (x1 * R.to(d)).sum().backward()
    for name, a, b in zip(["dx_src", "dW1", "db1", "dW2", "db2", "dWk"], dl, leaves):
        check(name, a.grad, b.grad, 2e-4)


@pytest.mark.parametrize("kind", ["star_in", "star_out", "sparse_sources", "chain_of_hubs"])
def test_edge_conv_ragged(kind):
    """Ragged graphs for the 16-row kernels (edge_conv16.hip): a destination with more in-edges than one 64-edge metadata batch, a source
    with more out-edges than one batch, long runs of nodes without edges (padded points) on either side, hubs that straddle the
    per-wave node chunks.  Forward, d x_src and the five weight gradients against the oracle (ragged / empty inputs)."""
    from geometry_rl_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(7)
    if kind == "star_in":          # 1100 destinations (> 512 forward tiles), destination 3 has 300 in-edges, the others one or none
        n_src, n_dst = 400, 1100
        src = torch.cat([torch.randint(0, n_src, (300,), generator=g), torch.randint(0, n_src, (600,), generator=g)])
        dst = torch.cat([torch.full((300,), 3), torch.randperm(n_dst, generator=g)[:600]])
    elif kind == "star_out":       # source 17 feeds 200 destinations; most sources have no out-edge at all
        n_src, n_dst = 1500, 1200
        src = torch.cat([torch.full((200,), 17), torch.randint(0, 40, (300,), generator=g) * 37])
        dst = torch.cat([torch.randperm(n_dst, generator=g)[:200], torch.randint(0, n_dst, (300,), generator=g)])
    elif kind == "sparse_sources":  # edges only among every 29th node: runs of 28 nodes without edges on both sides
        n_src = n_dst = 2900
        idx = torch.arange(0, 2900, 29)
        src = idx[torch.randint(0, 100, (700,), generator=g)]
        dst = idx[torch.randint(0, 100, (700,), generator=g)]
    else:                          # hubs of 70 / 130 / 65 in- and out-edges at nodes 15, 16, 31 (chunk boundaries for 16-node chunks)
        n_src = n_dst = 1300
        parts_s, parts_d = [], []
        for hub, deg in ((15, 70), (16, 130), (31, 65)):
            parts_s += [torch.full((deg,), hub), torch.randint(0, n_src, (deg,), generator=g)]
            parts_d += [torch.randint(0, n_dst, (deg,), generator=g), torch.full((deg,), hub)]
        src, dst = torch.cat(parts_s), torch.cat(parts_d)
    ei = torch.stack([src, dst])
    grid = eq.make_grid(3, 16, True)
    grid3 = F.pad(grid, (0, 3 - grid.shape[1]))
    x_src = torch.randn(n_src, 16, 64, generator=g)
    pos_s, pos_d = torch.rand(n_src, 3, generator=g) * 2 - 1, torch.rand(n_dst, 3, generator=g) * 2 - 1
    w1, b1, w2, b2, wk = params(g, [(64, 14), (64,), (64, 64), (64,), (64, 64)])
    R = torch.randn(n_dst, 16, 64, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x_src, w1, b1, w2, b2, wk)]
    xs, W1, B1, W2, B2, WK = leaves
    P = {"b.1.weight": W1, "b.1.bias": B1, "b.3.weight": W2, "b.3.bias": B2}
    kb = eq.basis_mlp(eq.spatial_invariants(grid, pos_s[ei[0]], pos_d[ei[1]]), P, "b")
    x1_ref = eq.scatter_sum(F.linear(kb, WK) * xs[ei[0]], ei[1], n_dst)
    (x1_ref * R).sum().backward()
    es = ops.build_edge_set(ei.to(d), n_src, n_dst)
    dl = [t.clone().to(d).requires_grad_(True) for t in (x_src, w1, b1, w2, b2, wk)]
    x1 = ops.EdgeConv.apply(dl[0], pos_s.to(d), pos_d.to(d), grid3.to(d), dl[1], dl[2], dl[3], dl[4], dl[5], es, 3)
    check("x1", x1, x1_ref)
    (x1 * R.to(d)).sum().backward()
    for name, a, b in zip(["dx_src", "dW1", "db1", "dW2", "db2", "dWk"], dl, leaves):
        check(name, a.grad, b.grad, 2e-4)


@pytest.mark.parametrize("n", [1, 7, 130, 700, 1601])   # 700 / 1601 nodes: 2-3 / 6-7 chunks per workgroup of the pipelined backward
def test_node_mlp(n):
    from geometry_rl_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(n)
    x2, xd, prev = (torch.randn(n, 16, 64, generator=g) for _ in range(3))
    gam, bet = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    w3, b3, w4, b4 = params(g, [(256, 64), (256,), (64, 256), (64,)])
    R = torch.randn(n, 16, 64, generator=g)
    for use_prev in (False, True):
        leaves = [t.clone().requires_grad_(True) for t in (x2, xd, gam, bet, w3, b3, w4, b4, prev)]
        X2, XD, G, Bt, W3, B3, W4, B4, PV = leaves
        h = F.layer_norm(X2, (64,), G, Bt, 1e-5)
        ref = XD + F.linear(F.gelu(F.linear(h, W3, B3)), W4, B4)
        if use_prev:
            ref = ref + PV
        (ref * R).sum().backward()
        dl = [t.clone().to(d).requires_grad_(True) for t in (x2, xd, gam, bet, w3, b3, w4, b4, prev)]
        out = ops.NodeMLP.apply(*dl[:8], dl[8] if use_prev else None)
        check(f"node_mlp out prev={use_prev}", out, ref)
        (out * R.to(d)).sum().backward()
        names = ["dx2", "dx_dst", "dgamma", "dbeta", "dW3", "db3", "dW4", "db4", "dprev"]
        for name, a, b in zip(names, dl, leaves):
            if name == "dprev" and not use_prev:
                continue
            check(name, a.grad, b.grad, 2e-4)


# node counts around the kernels' batching: fewer nodes than one batch of four, exact batches, a partial last batch, several batches per
# workgroup (9001), and more than 16 nodes per wave of the lift kernels' 4096 waves (70001: the staged inputs are refilled)
@pytest.mark.parametrize("n", [1, 3, 4, 5, 77, 9001, 70001])
def test_fiber_conv_and_lift(n):
    from geometry_rl_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(n)
    x1 = torch.randn(n, 16, 64, generator=g)
    fk = torch.randn(16, 16, 64, generator=g)
    bias = torch.randn(64, generator=g)
    R = torch.randn(n, 16, 64, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x1, fk, bias)]
    ref = torch.einsum("boc,opc->bpc", leaves[0], leaves[1]) / 16 + leaves[2]
    (ref * R).sum().backward()
    dl = [t.clone().to(d).requires_grad_(True) for t in (x1, fk, bias)]
    out = ops.FiberConv.apply(*dl)
    check("x2", out, ref)
    (out * R.to(d)).sum().backward()
    for name, a, b in zip(["dx1", "dfk", "dbias"], dl, leaves):
        check(name, a.grad, b.grad, 2e-4)

    for dim in (3, 2):
        grid = eq.make_grid(dim, 16)
        grid3 = F.pad(grid, (0, 3 - grid.shape[1]))
        scal = torch.zeros(n, 3)
        scal[:, n % 3] = 1
        vec = torch.randn(n, 4, 3, generator=g)
        w = torch.randn(64, 7, generator=g)
        wl = w.clone().requires_grad_(True)
        ref = F.linear(eq.lift_features(scal, vec.reshape(n, -1), grid, dim), wl)
        (ref * R).sum().backward()
        wd = w.clone().to(d).requires_grad_(True)
        out = ops.LiftEncode.apply(scal.to(d), vec.to(d), grid3.to(d), wd)
        check(f"lift dim{dim}", out, ref)
        (out * R.to(d)).sum().backward()
        check(f"lift dW dim{dim}", wd.grad, wl.grad, 2e-4)


# The plain-bf16 build of the same kernels (entry points *_bf16: bf16 storage, fp32 arithmetic -- round 5 gave them their own inner loops:
# packed backward batch, bursts of raw quads in the lift backward, raw prefetch registers).  Inputs are rounded to bf16 FIRST, so the fp32
# torch reference sees exactly the values the kernels load: what is left is the summation order (weight gradients: fp32 partial sums,
# 2e-4 of the tensor's largest entry as for the fp32 build) and ONE rounding of each stored latent to nearest bf16 (2^-8 relative).
@pytest.mark.parametrize("n", [1, 3, 4, 5, 77, 9001, 70001])
def test_fiber_conv_and_lift_bf16_build(n):
    from geometry_rl_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(100 + n)
    r16 = lambda t: t.to(torch.bfloat16)
    x1 = r16(torch.randn(n, 16, 64, generator=g))
    fk = torch.randn(16, 16, 64, generator=g)
    bias = torch.randn(64, generator=g)
    R = r16(torch.randn(n, 16, 64, generator=g))
    leaves = [x1.float().requires_grad_(True), fk.clone().requires_grad_(True), bias.clone().requires_grad_(True)]
    ref = torch.einsum("boc,opc->bpc", leaves[0], leaves[1]) / 16 + leaves[2]
    (ref * R.float()).sum().backward()
    dl = [x1.to(d).requires_grad_(True), fk.clone().to(d).requires_grad_(True), bias.clone().to(d).requires_grad_(True)]
    out = ops.FiberConv.apply(*dl, "_bf16")
    assert out.dtype == torch.bfloat16

    def close16(name, got, want):   # a stored latent: nearest bf16 of an fp32 value that itself differs in the last fp32 bits
        err = (got.float().cpu() - want).abs()
        tol = 2.0 ** -8 * want.abs() + 1e-5 * float(want.abs().max())
        assert bool((err <= tol).all()), (name, float((err - tol).max()))

    close16("x2", out.detach(), ref.detach())
    (out.float() * R.to(d).float()).sum().backward()   # d out = R (bf16-exact), handed to the kernel as bf16
    close16("dx1", dl[0].grad, leaves[0].grad)
    check("dfk", dl[1].grad, leaves[1].grad, 2e-4)
    check("dbias", dl[2].grad, leaves[2].grad, 2e-4)

    for dim in (3, 2):
        grid = eq.make_grid(dim, 16)
        grid3 = F.pad(grid, (0, 3 - grid.shape[1]))
        scal = torch.zeros(n, 3)
        scal[:, n % 3] = 1
        vec = torch.randn(n, 4, 3, generator=g)
        w = torch.randn(64, 7, generator=g)
        wl = w.clone().requires_grad_(True)
        ref = F.linear(eq.lift_features(scal, vec.reshape(n, -1), grid, dim), wl)
        (ref * R.float()).sum().backward()
        wd = w.clone().to(d).requires_grad_(True)
        out = ops.LiftEncode.apply(scal.to(d), vec.to(d), grid3.to(d), wd, "_bf16")
        close16(f"lift dim{dim}", out.detach(), ref.detach())
        (out.float() * R.to(d).float()).sum().backward()
        check(f"lift dW dim{dim}", wd.grad, wl.grad, 2e-4)


def test_reduce_partials_multi_all_paths():
    """Gradient folding in one launch: float4 path (aligned slabs), scalar path (odd lengths / offsets), several slabs feeding one
    destination, accumulation into non-zero destinations, row counts around the unroll depths."""
    import ctypes
    from geometry_rl_amd import hip
    d = dev()
    g = torch.Generator().manual_seed(3)
    slabs = [torch.randn(r, ld, generator=g).to(d) for r, ld in ((1024, 9216), (256, 9216), (37, 4353), (600, 448), (1, 64), (513, 132))]
    # (slab, start, len, destination key)
    segs = [(0, 960, 4096, "w2"), (1, 960, 4096, "w2"), (0, 0, 896, "w1"), (1, 0, 896, "w1"), (0, 5120, 4096, "wk0"), (1, 5120, 4096, "wk1"),
            (2, 0, 4096, "a"), (2, 4096, 64, "b"), (2, 4224, 1, "c"), (2, 4225, 64, "odd_start"), (3, 0, 448, "enc"), (4, 0, 64, "one_row"),
            (5, 3, 127, "odd_len"), (5, 4, 128, "aligned_in_odd_ld")]
    dsts, ref = {}, {}
    for _, _, ln, k in segs:
        if k not in dsts:
            dsts[k] = torch.randn(ln, generator=g).to(d)
            ref[k] = dsts[k].double().clone()
    for si, st, ln, k in segs:
        ref[k] += slabs[si][:, st:st + ln].double().sum(0)
    n = len(segs)
    hip.call("grl_reduce_partials_multi", n, (ctypes.c_void_p * n)(*[slabs[s].data_ptr() for s, _, _, _ in segs]),
             (ctypes.c_int * n)(*[slabs[s].shape[0] for s, _, _, _ in segs]), (ctypes.c_int * n)(*[slabs[s].shape[1] for s, _, _, _ in segs]),
             (ctypes.c_int * n)(*[st for _, st, _, _ in segs]), (ctypes.c_int * n)(*[ln for _, _, ln, _ in segs]),
             (ctypes.c_void_p * n)(*[dsts[k].data_ptr() for _, _, _, k in segs]))
    torch.cuda.synchronize()
    for k in dsts:
        err = (dsts[k].double() - ref[k]).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref[k].abs().max().item()), (k, err)


@pytest.mark.parametrize("B,n,d", [(1, 3, 15), (7, 35, 15), (130, 241, 12), (1030, 9, 7),
                                   # 64 or more rows per sample: the four-rows-per-access kernels (critic_ops.hip, QUAD_MIN_N) -- widest / narrowest
                                   # input, row counts that are no multiple of four, fewer samples than waves, more workgroups than slots
                                   (3, 64, 16), (5, 67, 1), (9, 130, 13), (1100, 65, 3), (2, 257, 15)])
def test_deepsets_critic_shapes(B, n, d):
    """The six critic kernels (slot statistics, register-resident weights, 16-wave row kernels) against the oracle's DeepSets over
    batch / set / feature sizes around the kernels' tiling (fewer samples than waves, more workgroups than statistic slots ...)."""
    from geometry_rl_amd import ops
    from oracle import graph as ogr
    dv = dev()
    P = ogr.init_critic_params(d, seed=5)
    g = torch.Generator().manual_seed(B + n)
    x = torch.randn(B, n, d, generator=g)
    R = torch.randn(B, generator=g)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = ogr.value_forward(Pg, x).reshape(B)
    (ref * R).sum().backward()
    order = ["gnn.mlp_inner.lins.0.weight", "gnn.mlp_inner.lins.0.bias", "gnn.mlp_inner.norms.0.weight", "gnn.mlp_inner.norms.0.bias",
             "gnn.mlp_inner.lins.1.weight", "gnn.mlp_inner.lins.1.bias", "gnn.mlp_outer.lins.0.weight", "gnn.mlp_outer.lins.0.bias",
             "gnn.mlp_outer.norms.0.weight", "gnn.mlp_outer.norms.0.bias", "gnn.mlp_outer.lins.1.weight", "gnn.mlp_outer.lins.1.bias",
             "final.weight", "final.bias"]
    leaves = [P[k].clone().to(dv).requires_grad_(True) for k in order]
    val = ops.DeepSetsValue.apply(x.to(dv), *leaves, None)
    (val * R.to(dv)).sum().backward()
    scale = max(1.0, float(ref.detach().abs().max()))
    assert float((val.detach().cpu() - ref.detach()).abs().max()) <= 2e-5 * scale
    for k, t in zip(order, leaves):
        gr = Pg[k].grad
        err = float((t.grad.cpu() - gr).abs().max())
        assert err <= 1e-4 * max(1.0, float(gr.abs().max())), (k, err, float(gr.abs().max()))


def test_kernel_prof_stamp_mode_times_a_replayed_launch():
    """grl_prof_enable(2): wall-clock stamp kernels around a launch are ordinary graph nodes, so the duration of a REPLAYED launch can
    be read back (bench.py's `roofline.replayed_launches`).  Here: the node-MLP backward recorded into a hipGraph, replayed three
    times; every replay yields a fresh, plausible duration that agrees with HIP events around the eager launch."""
    from geometry_rl_amd import hip
    dv = dev()
    g = torch.Generator().manual_seed(3)
    n = 4096
    x2, dout = (torch.randn(n, 16, 64, generator=g).to(dv) for _ in range(2))
    w3, b3, w4, b4 = (torch.randn(s, generator=g).mul(0.1).to(dv) for s in [(256, 64), (256,), (64, 256), (64,)])
    gam, bet = torch.ones(64, device=dv), torch.zeros(64, device=dv)
    rows = n * 16
    blocks = hip.query("grl_node_mlp_bwd_blocks", rows)
    partial = torch.empty(blocks + 1, hip.query("grl_node_mlp_partial_size"), device=dv)
    dx2 = torch.empty_like(x2)
    run = lambda: hip.call("grl_node_mlp_bwd", x2, dout, w3, b3, w4, b4, gam, bet, dx2, partial, rows)
    for _ in range(2):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    eager_ms = e0.elapsed_time(e1)
    hip.kernel_prof_enable(2)
    try:
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            run()
        torch.cuda.current_stream().wait_stream(side)
        seen = []
        for _ in range(3):
            graph.replay()
            torch.cuda.synchronize()
            summ = hip.kernel_prof_summary()
            assert list(summ) == ["node_mlp_bwd16_kernel"] and summ["node_mlp_bwd16_kernel"][0] == 1, summ
            seen.append(summ["node_mlp_bwd16_kernel"][1])
    finally:
        hip.kernel_prof_enable(False)
    print("eager (events)", eager_ms, "ms; replayed (stamps)", seen)
    for ms in seen:
        assert 0.3 * eager_ms < ms < 2.0 * eager_ms + 0.05, (ms, eager_ms)
    assert hip.kernel_prof_summary() == {}


@pytest.mark.parametrize("S,V", [(1, 7), (2, 1), (5, 3), (8, 0)])
def test_lift_feature_splits(S, V):
    """Lift + encoder with other scalar / vector feature counts than the bench configs' (the kernels fetch a node's S + 3 V inputs with one
    load and clamp the unused slots: widest vector block, a single vector, S + V = 8 = the kernels' maximum, no vectors at all)."""
    from geometry_rl_amd import ops
    d = dev()
    n = 301
    g = torch.Generator().manual_seed(10 * S + V)
    grid = eq.make_grid(3, 16)
    scal = torch.randn(n, S, generator=g)
    vec = torch.randn(n, V, 3, generator=g)
    w = torch.randn(64, S + V, generator=g)
    R = torch.randn(n, 16, 64, generator=g)
    wl = w.clone().requires_grad_(True)
    # x[n,o,c] = sum_s scal[n,s] W[c,s] + sum_v (vec[n,v,:] . grid[o,:]) W[c,S+v]     (hepi.py:136-143, to_from_sphere.py:4-9)
    feat = torch.cat([scal[:, None, :].expand(n, 16, S), torch.einsum("nvd,od->nov", vec, grid)], -1)
    ref = F.linear(feat, wl)
    (ref * R).sum().backward()
    wd = w.clone().to(d).requires_grad_(True)
    out = ops.LiftEncode.apply(scal.to(d), vec.to(d), grid.to(d), wd)
    check(f"lift S{S} V{V}", out, ref)
    (out * R.to(d)).sum().backward()
    check(f"lift dW S{S} V{V}", wd.grad, wl.grad, 2e-4)


@pytest.mark.parametrize("n", [1, 3, 5, 130, 4099])
def test_readout_head(n):
    """Decoder + orientation pooling + contextual std head (ops.Readout) against the oracle's readout / std_head, forward and backward, over
    node counts around the backward kernel's tiling (four waves per workgroup, one node per wave: fewer nodes than waves, a partial last
    workgroup, more nodes than the 1024 x 4 resident waves)."""
    from geometry_rl_amd import ops
    from oracle import trpl as otr
    d = dev()
    od = ov = 2
    g = torch.Generator().manual_seed(n)
    grid = eq.make_grid(3, 16)
    lat = torch.randn(n, 16, 64, generator=g)
    wd, bd = torch.randn(od + ov, 64, generator=g) * 0.2, torch.randn(od + ov, generator=g) * 0.1
    ws, bs = torch.randn(3 * ov, 64, generator=g) * 0.2, torch.randn(3 * ov, generator=g) * 0.1
    Rm, Rs, Rh = torch.randn(n, ov, 3, generator=g), torch.randn(n, 3 * ov, generator=g), torch.randn(n, 64, generator=g)
    init_std, min_std = 1.0, 1e-5
    shift = float(otr.inverse_softplus(torch.tensor(init_std - min_std)))
    ref_leaves = [t.clone().requires_grad_(True) for t in (lat, wd, bd, ws, bs)]
    mean_r, hid_r = eq.readout(ref_leaves[0], ref_leaves[1], ref_leaves[2], grid, 3, od, ov)
    sig_r = otr.std_head(hid_r, ref_leaves[3], ref_leaves[4], init_std, min_std, n)
    ((mean_r.reshape(n, ov, 3) * Rm).sum() + (sig_r * Rs).sum() + (hid_r * Rh).sum()).backward()
    dl = [t.clone().to(d).requires_grad_(True) for t in (lat, wd, bd, ws, bs)]
    mean, sigma, hidden = ops.Readout.apply(dl[0], grid.to(d), dl[1], dl[2], dl[3], dl[4], shift, min_std, od, ov)
    check("mean", mean.reshape(-1, 3), mean_r.reshape(-1, 3))
    check("sigma", sigma, sig_r)
    check("hidden", hidden, hid_r)
    ((mean * Rm.to(d)).sum() + (sigma * Rs.to(d)).sum() + (hidden * Rh.to(d)).sum()).backward()
    for name, a, b in zip(["dlat", "dWd", "dbd", "dWs", "dbs"], dl, ref_leaves):
        check(name, a.grad, b.grad, 2e-4)


def test_fixed_order_sums_are_repeatable():
    """grl_adv_stats and grl_clip_coef: one workgroup, fixed summation order -- the fp64 sums match an fp64 reference and two launches give the
    same BITS, on views of any alignment and length (the flat gradient is clipped per network: views that start anywhere)."""
    from geometry_rl_amd import hip
    d = dev()
    g = torch.Generator().manual_seed(11)
    base = torch.randn(300000, generator=g).to(d)
    for lo, n in ((0, 149000), (1, 7), (3, 150001), (2, 1), (5, 280013), (0, 4), (7, 4096)):
        v = base[lo:lo + n]
        outs = []
        for _ in range(2):
            sq = torch.zeros(1, device=d, dtype=torch.float64)
            coef = torch.empty(1, device=d)
            hip.call("grl_clip_coef", v, n, 0.5, sq, coef)
            st = torch.zeros(2, device=d, dtype=torch.float64)
            hip.call("grl_adv_stats", v.contiguous(), st, n)
            outs.append((float(sq), float(coef), float(st[0]), float(st[1])))
        assert outs[0] == outs[1], (lo, n, outs)
        ref2 = float((v.double() ** 2).sum())
        assert abs(outs[0][0] - ref2) <= 1e-13 * ref2 and abs(outs[0][3] - ref2) <= 1e-13 * ref2, (lo, n)
        assert abs(outs[0][2] - float(v.double().sum())) <= 1e-10 * max(1.0, n ** 0.5), (lo, n)
        assert abs(outs[0][1] - min(1.0, 0.5 / (ref2 ** 0.5 + 1e-6))) <= 1e-6
