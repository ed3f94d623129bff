"""torch.autograd.Function wrappers around the C-ABI HIP kernels (include/grl_hip.h).

Only plumbing lives here: argument checks, workspace allocation with torch's caching allocator, and wiring each
forward/backward kernel pair into autograd so the modules in ``geometry_rl_amd.modules`` behave like ordinary
``nn.Module``s.  No arithmetic of the hot path is done in Python/PyTorch.
"""
from dataclasses import dataclass
from typing import Optional


import torch

from . import hip


@dataclass
class EdgeSet:
    """One edge type of the batched graph as a destination-sorted CSR (int32, device resident); forward and backward
    kernels walk the same arrays (see csrc/edge_conv.hip)."""
    n_src: int
    n_dst: int
    n_edges: int
    rowptr_d: torch.Tensor
    src_d: torch.Tensor
    dst_d: torch.Tensor
    rowptr_s: torch.Tensor  # the same edges as a source-sorted CSR (the backward sums d x_src per source node in registers)
    src_s: torch.Tensor
    dst_s: torch.Tensor
    s2d: Optional[torch.Tensor] = None  # row of the i-th source-sorted edge in the destination-sorted order (attention aggregation)
    split_s: Optional[torch.Tensor] = None  # [4 * grl_edge_bwd_blocks(E) + 1] node boundaries of an edge-balanced partition of the
    #                                         source-sorted CSR over the backward's wave slots (None: the round-robin deal is balanced)
    split_d: Optional[torch.Tensor] = None  # the same for the destination-sorted CSR and the forward's wave slots ([n_slots + 1])


def build_edge_set(edge_index: torch.Tensor, n_src: int, n_dst: int, split_s: Optional[torch.Tensor] = None) -> EdgeSet:
    """edge_index [2,E] (row 0 = source, row 1 = destination) -> CSR by destination.  Runs once per cached topology, never under stream
    capture: it synchronises with the host (sort keys, the balance check of the wave partitions)."""
    src, dst = edge_index[0].long(), edge_index[1].long()
    dev = edge_index.device
    if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("build_edge_set synchronises with the host: build the topology before the step is recorded "
                           "(HyperData.check_topology / the first, eager step do)")

    def csr(anchor, other, n_anchor):
        order = torch.argsort(anchor * (int(other.max().item()) + 1 if other.numel() else 1) + other)
        a, o = anchor[order], other[order]
        rowptr = torch.zeros(n_anchor + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(a, minlength=n_anchor), 0)
        return rowptr.int().contiguous(), a.int().contiguous(), o.int().contiguous(), order

    rp_d, dst_d, src_d, order_d = csr(dst, src, n_dst)
    rp_s, src_s, dst_s, order_s = csr(src, dst, n_src)
    pos_d = torch.empty_like(order_d)
    pos_d[order_d] = torch.arange(order_d.numel(), device=dev)      # original edge id -> destination-sorted position
    s2d = pos_d[order_s].int().contiguous()                         # source-sorted position -> destination-sorted position
    # Edge-balanced wave partitions for the 16-row edge kernels (csrc/edge_conv16.hip): wave slot s walks the anchor nodes split[s] ..
    # split[s + 1], which carry ~E / slots edges.  Built only where the kernel's round-robin deal of node chunks leaves the slowest wave
    # > 4 % above the mean: in the backward the OUT-degrees of a kNN graph vary (rigid HEPi, every batch size: -2.5 % on the 4096-frame
    # step, -5.5 % at 1024 frames); in the forward 13 108 five-node chunks over 3 072 slots leave some waves five chunks, others four.
    # Functions of the topology alone: results stay reproducible.
    E = int(src.numel())

    def balanced_split(rowptr, n_anchor, slots, npw, tol):
        """Node boundaries [slots + 1] with ~E / slots edges per slot, or None where the slowest wave slot of the kernel's round-robin deal
        of npw-node chunks over ``slots`` slots carries at most ``tol`` x the mean load (uniform-degree edge sets, whole frames per chunk)."""
        rp = rowptr.long()
        n_chunks = (n_anchor + npw - 1) // npw
        cb = (torch.arange(n_chunks + 1, device=dev, dtype=torch.int64) * npw).clamp_(max=n_anchor)
        load = torch.zeros(slots, device=dev, dtype=torch.int64).scatter_add_(0, torch.arange(n_chunks, device=dev) % slots, rp[cb[1:]] - rp[cb[:-1]])
        if float(load.max()) * slots / E <= tol:
            return None
        cum = rp.double()   # cost of a node = its edges (one pass each)
        targets = torch.arange(slots + 1, device=dev, dtype=torch.float64) * (float(cum[-1]) / slots)
        split = torch.searchsorted(cum, targets).clamp_(max=n_anchor)
        split[0], split[-1] = 0, n_anchor
        return split.int().contiguous()

    given_s, split_s, split_d = split_s, None, None   # (``split_s`` given: the caller numbered the nodes for it -- graph.balanced_node_order)
    if E > 0 and dev.type == "cuda":   # (the launch shapes are the library's: asked for, not copied -- ADVICE r3; CPU tensors: no partitions)
        # backward (csrc/edge_conv16.hip grl_edge_bwd16_launch): 4 waves x grl_edge_bwd_blocks(E) workgroups, chunks of
        # grl_edge_bwd_chunk_nodes(n_src) nodes dealt round-robin (one wave per SIMD: a wave that finishes early leaves its SIMD idle --
        # every per cent of imbalance is a per cent of the launch)
        if given_s is not None:
            if given_s.numel() != 4 * hip.query("grl_edge_bwd_blocks", E) + 1 or int(given_s[-1]) != n_src:
                raise ValueError("build_edge_set: split_s must hold 4 * grl_edge_bwd_blocks(E) + 1 node boundaries ending at n_src")
            split_s = given_s.int().contiguous()
        else:
            split_s = balanced_split(rp_s, n_src, 4 * hip.query("grl_edge_bwd_blocks", E), hip.query("grl_edge_bwd_chunk_nodes", n_src), 1.04)
        # forward (grl_edge16_launch): grl_edge_fwd_slots(n_dst) wave slots (0: the launch takes the one-workgroup-per-tile kernel of small
        # graphs, which ignores partitions), chunks of grl_edge_fwd_chunk_nodes(n_dst) nodes
        slots_d = hip.query("grl_edge_fwd_slots", n_dst)
        if slots_d > 0:
            # three waves share a SIMD here: the waves that are left when the others finish run faster, so a moderately uneven deal heals
            # itself (rigid HEPi, 17 % uneven: the partition costs 0.9 % of the step) -- only grossly uneven graphs are partitioned (the
            # merged EMPN graph: actuator nodes with 17 in-edges behind object nodes with 3: -2 %)
            split_d = balanced_split(rp_d, n_dst, slots_d, hip.query("grl_edge_fwd_chunk_nodes", n_dst), 1.25)
    return EdgeSet(n_src, n_dst, E, rp_d, src_d, dst_d, rp_s, src_s, dst_s, s2d, split_s, split_d)


def _reduce(partial: torch.Tensor, out: torch.Tensor):
    hip.call("grl_reduce_partials", partial, out, partial.shape[0], out.numel())


# Deferred folding: while a list is installed here (PolicyUpdater does, around the backward), folds whose destinations are all
# existing leaf ``.grad`` buffers are queued and executed by ONE launch (flush_deferred_grads) instead of one launch each.
SPLIT_BACKWARD = True   # edge-balanced wave partition in the fused edge backward (EdgeSet.split_s); module attributes, not environment
SPLIT_FORWARD = True    # ... and in the forward (EdgeSet.split_d): tools flip them in-process for A/B runs
DEFERRED = None


def _launch_folds(jobs, overwrite=False):
    """One launch per <= 64 slabs; every slab of a destination goes into the SAME launch (the kernel sums a destination's slabs in one
    workgroup, in the order given) -- with ``overwrite`` that is a requirement, not just the cheaper form."""
    import ctypes
    order, by_dst = [], {}
    for j in jobs:                       # group by destination, first-appearance order (the summation order of a destination's slabs)
        k = j[3].data_ptr()
        if k not in by_dst:
            by_dst[k] = []
            order.append(k)
        by_dst[k].append(j)
    batches, cur = [], []
    for k in order:
        grp = by_dst[k]
        if len(grp) > 64:
            raise RuntimeError("more than 64 partial slabs feed one gradient")
        if len(cur) + len(grp) > 64:
            batches.append(cur)
            cur = []
        cur += grp
    if cur:
        batches.append(cur)
    for part in batches:
        n = len(part)
        hip.call("grl_reduce_partials_multi_ow", n, (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in part]),
                 (ctypes.c_int * n)(*[j[0].shape[0] for j in part]), (ctypes.c_int * n)(*[j[0].shape[1] for j in part]),
                 (ctypes.c_int * n)(*[j[1] for j in part]), (ctypes.c_int * n)(*[j[2] for j in part]),
                 (ctypes.c_void_p * n)(*[j[3].data_ptr() for j in part]), 1 if overwrite else 0)


def fold_adam_report(overwrite, adam=None, report=None, signal=None):
    """The queued folds, the Adam update of the parameters they feed and the reported loss values in ONE launch
    (grl_fold_adam_report).  ``adam``: dict(grads, params, exp_avg, exp_avg_sq, lr_dev, betas, eps, step_dev) or None; ``report``:
    dict(slots, batch, sums, maxes, ent_coef, out14) or None.  Returns False (nothing launched, queue untouched) when the queue does not fit
    one launch -- the caller then takes the separate launches."""
    import ctypes
    global DEFERRED
    jobs = DEFERRED or []
    if len(jobs) > 64:
        return False
    by_dst = {}
    for j in jobs:
        by_dst.setdefault(j[3].data_ptr(), []).append(j)
    part = [j for grp in by_dst.values() for j in grp]
    n = len(part)
    a, r = adam or {}, report or {}
    if adam is not None:   # every destination must lie inside the flat gradient buffer the optimizer state is parallel to
        lo, hi = a["grads"].data_ptr(), a["grads"].data_ptr() + 4 * a["grads"].numel()
        if any(not (lo <= j[3].data_ptr() and j[3].data_ptr() + 4 * j[2] <= hi) for j in part):
            return False
    sig = list(signal) if signal is not None else [None, None]   # (flag_dst, flag_src): a lane signal written when the launch starts
    hip.call("grl_fold_adam_report_sig", n, (ctypes.c_void_p * max(n, 1))(*[j[0].data_ptr() for j in part]),
             (ctypes.c_int * max(n, 1))(*[j[0].shape[0] for j in part]), (ctypes.c_int * max(n, 1))(*[j[0].shape[1] for j in part]),
             (ctypes.c_int * max(n, 1))(*[j[1] for j in part]), (ctypes.c_int * max(n, 1))(*[j[2] for j in part]),
             (ctypes.c_void_p * max(n, 1))(*[j[3].data_ptr() for j in part]), 1 if overwrite else 0, 1 if adam is not None else 0,
             a.get("grads"), a.get("params"), a.get("exp_avg"), a.get("exp_avg_sq"), a.get("lr_dev"),
             float(a["betas"][0]) if adam else 0.0, float(a["betas"][1]) if adam else 0.0, float(a["eps"]) if adam else 0.0,
             a.get("step_dev"), r.get("slots"), int(r.get("batch", 0)), r.get("sums"), r.get("maxes"), float(r.get("ent_coef", 0.0)),
             r.get("out14"), *sig)
    if DEFERRED is not None:
        DEFERRED = []
    return True


def fold_record_pairs(overwrite, slots, batch, rank, world, region) -> bool:
    """Data parallel: the queued folds AND this rank's loss record (as (hi, lo) float pairs in front of the flat gradient) in ONE launch
    (grl_fold_record_pairs) -- both only feed the lane's all-reduce.  False (nothing launched) when the queue does not fit one launch."""
    import ctypes
    global DEFERRED
    jobs = DEFERRED or []
    if len(jobs) > 64:
        return False
    by_dst = {}
    for j in jobs:
        by_dst.setdefault(j[3].data_ptr(), []).append(j)
    part = [j for grp in by_dst.values() for j in grp]
    n = len(part)
    hip.call("grl_fold_record_pairs", n, (ctypes.c_void_p * max(n, 1))(*[j[0].data_ptr() for j in part]),
             (ctypes.c_int * max(n, 1))(*[j[0].shape[0] for j in part]), (ctypes.c_int * max(n, 1))(*[j[0].shape[1] for j in part]),
             (ctypes.c_int * max(n, 1))(*[j[1] for j in part]), (ctypes.c_int * max(n, 1))(*[j[2] for j in part]),
             (ctypes.c_void_p * max(n, 1))(*[j[3].data_ptr() for j in part]), 1 if overwrite else 0, slots, int(batch), int(rank), int(world), region)
    if DEFERRED is not None:
        DEFERRED = []
    return True


def flush_deferred_grads(overwrite=False, only=None):
    """Fold every queued slab into its leaf gradient.  ``overwrite``: the gradients are WRITTEN (the caller keeps no zeroed buffer; every
    leaf gradient of the pass must then come through this queue -- PolicyUpdater checks that).  ``only``: a predicate on the destination
    tensor -- fold just those jobs now and leave the others queued (the critic's lane folds its own gradients)."""
    global DEFERRED
    jobs = DEFERRED or []
    keep = []
    if only is not None:
        keep = [j for j in jobs if not only(j[3])]
        jobs = [j for j in jobs if only(j[3])]
    if DEFERRED is not None:
        DEFERRED = keep
    if jobs:
        _launch_folds(jobs, overwrite)


def _emit_grads(partial: torch.Tensor, segments):
    """Fold per-workgroup partial rows into gradients with ONE launch.  ``segments`` = [(start, length, shape, tensor)]:
    when ``tensor`` is a leaf whose ``.grad`` buffer already exists (PolicyUpdater keeps every parameter's grad as a view of
    one flat buffer) the sum is accumulated in place and ``None`` is handed to autograd (no AccumulateGrad add kernel);
    otherwise a fresh gradient tensor is returned.  While a DEFERRED queue is installed the leaf segments are only queued (one launch
    for all of them at the end of the backward pass) and the fresh ones are produced at once."""
    import ctypes
    dev = partial.device
    outs, dsts, starts, lens, fresh_flags = [], [], [], [], []
    skip = set()
    for i_, (start, length, shape, t) in enumerate(segments):
        if t is not None and t.is_leaf and not t.requires_grad:
            skip.add(i_)   # a constant (e.g. the zero weight of a state-independent std head): nobody reads this gradient
        g = getattr(t, "grad", None) if (t is not None and t.is_leaf) else None
        if g is not None and g.is_contiguous() and g.dtype == torch.float32:
            outs.append(None)
            dsts.append(g)
            fresh_flags.append(0)
        else:
            fresh = torch.empty(shape, device=dev, dtype=torch.float32)  # written (not accumulated) by the reduction
            outs.append(fresh)
            dsts.append(fresh)
            fresh_flags.append(1)
        starts.append(start)
        lens.append(length)
    now = [i for i in range(len(segments)) if i not in skip]
    for i in skip:
        outs[i] = None
    if DEFERRED is not None:
        jobs = [(partial, st, ln, d) for i, (st, ln, d, f) in enumerate(zip(starts, lens, dsts, fresh_flags)) if not f and i not in skip]
        now = [i for i, f in enumerate(fresh_flags) if f and i not in skip]
        DEFERRED.extend(jobs)   # keeps `partial` alive until the flush
    for i0 in range(0, len(now), 8):
        idx = now[i0:i0 + 8]
        n = len(idx)
        mask = sum(fresh_flags[i] << k for k, i in enumerate(idx))
        hip.call("grl_reduce_partials_seg", partial, partial.shape[0], partial.shape[1], n,
                 (ctypes.c_void_p * n)(*[dsts[i].data_ptr() for i in idx]), (ctypes.c_int * n)(*[starts[i] for i in idx]),
                 (ctypes.c_int * n)(*[lens[i] for i in idx]), mask)
    return outs


# ---- merged launches of the recorded step (round 6; csrc/node_ops.hip step_head_kernel / lift_fiber_basis_bwd_kernel) ---------------------
import os as _os
# module attributes (tests flip them in-process; the A/B scripts through the environment): False = every role is its own launch, as in round 5
FUSE_HEAD = _os.environ.get("GRL_FUSE_HEAD", "1") == "1"
FUSE_TAIL_PRE = _os.environ.get("GRL_FUSE_TAIL_PRE", "1") == "1"
SIGNAL_IN_KERNEL = _os.environ.get("GRL_SIGNAL_IN_KERNEL", "1") == "1"   # False: lane signals as 4-byte copy launches of their own
HEAD = None            # a HeadLaunch while an actor forward that supports it is being issued (policy.GNNGaussianPolicyDiag.forward_diag)
TAIL_PRE = None        # a dict while PolicyUpdater issues the actor's backward: the first of {lift backward, fiber-basis backward} waits here
AFTER_FIBER_HOOK = None
PENDING_SIGNAL = None  # (flag_dst, flag_src) int32 tensors: the next FiberConv forward launch writes flag_dst[0] = flag_src[0] when it starts


class HeadLaunch:
    """The three mutually independent launches at the head of an actor forward -- node features (+ the optimizer's step count), fiber basis
    + fiber kernels, pre-split weight images -- collected and issued as ONE (grl_step_head).  Each producer hands over its argument set
    instead of launching (graph.HyperData.build_data, FiberKernels.forward, weight_images); ``launch`` must run before the first consumer
    (the lift reads the features)."""

    def __init__(self):
        self.feat = self.fiber = self.wimg = None
        self.keep = []

    def launch(self, prec: str = ""):
        import ctypes
        if self.feat is None and self.fiber is None and self.wimg is None:
            return
        words, n_desc, bump = self.feat if self.feat is not None else (None, 0, None)
        if self.fiber is not None:
            poly2, P, W, saved, fks = self.fiber
            n = len(W)
            fargs = [poly2, *P, (ctypes.c_void_p * n)(*[t.data_ptr() for t in W]), n, saved, (ctypes.c_void_p * n)(*[t.data_ptr() for t in fks])]
        else:
            fargs = [None, None, None, None, None, None, 0, None, None]
        if self.wimg is not None:
            kinds, ptrs, outs = self.wimg
            n_img = len(kinds)
            wargs = [n_img, (ctypes.c_int * n_img)(*kinds), (ctypes.c_void_p * (6 * n_img))(*ptrs), (ctypes.c_void_p * n_img)(*outs)]
        else:
            wargs = [0, None, None, None]
        hip.call("grl_step_head" + prec, words, n_desc, bump, *fargs, *wargs)
        self.feat = self.fiber = self.wimg = None


def _tail_pre_launch(lift, fiber):
    """lift = (prec, T, scal, vec, grid3, dxs, partial, ns, s, v) or None; fiber = (poly2, w2, W, saved, d, partial) or None: ONE launch."""
    import ctypes
    prec = lift[0] if lift is not None else ""
    if lift is not None:
        _, T, scal, vec, grid3, dxs, lpart, ns, s_, v_ = lift
        largs = [T, (ctypes.c_void_p * T)(*[a.data_ptr() for a in scal]), (ctypes.c_void_p * T)(*[b.data_ptr() for b in vec]), grid3,
                 (ctypes.c_void_p * T)(*[(d.data_ptr() if d is not None else 0) for d in dxs]), lpart, ns, s_, v_]
    else:
        largs = [0, None, None, None, None, None, None, 0, 0]
    if fiber is not None:
        poly2, w2, W, saved, d, fpart = fiber
        n = len(W)
        fargs = [poly2, w2, (ctypes.c_void_p * n)(*[t.data_ptr() for t in W]), n, saved,
                 (ctypes.c_void_p * n)(*[(g.data_ptr() if g is not None else 0) for g in d]), fpart]
    else:
        fargs = [None, None, None, 0, None, None, None]
    hip.call("grl_lift_fiber_basis_bwd" + prec, *largs, *fargs)


def _tail_pre_offer(kind, job):
    """Called by the two backward functions while TAIL_PRE is installed: the first one parks its launch, the second issues both as one."""
    other = "fiber" if kind == "lift" else "lift"
    if other in TAIL_PRE:
        pair = {kind: job, other: TAIL_PRE.pop(other)}
        _tail_pre_launch(pair["lift"], pair["fiber"])
    else:
        TAIL_PRE[kind] = job


def flush_tail_pre():
    """Issue whatever is still parked (a backward in which only one of the two roles ran)."""
    if TAIL_PRE:
        _tail_pre_launch(TAIL_PRE.pop("lift", None), TAIL_PRE.pop("fiber", None))


def _all_leaf_grads(tensors):
    return all(t is not None and t.is_leaf and getattr(t, "grad", None) is not None and t.grad.is_contiguous() and t.grad.dtype == torch.float32
               for t in tensors)


class WeightImages:
    """Pre-split weight images of ONE convolution block for one forward pass (csrc/grl_wimg.h, grl_weight_images): device byte buffers
    whose contents are the MFMA kernels' LDS structs.  ``e16``: edge chain for the 16-row kernels (forward prefix + the backward's
    transposes), ``e32``: edge chain of the few-tile 32-row forward, ``mlp_f`` / ``mlp_b``: ConvNeXt block forward / backward.  Any
    member may be None (the kernel then stages the weights itself).  Valid until the weights change (the optimizer step)."""
    __slots__ = ("e16", "e32", "mlp_f", "mlp_b")

    def __init__(self, e16=None, e32=None, mlp_f=None, mlp_b=None):
        self.e16, self.e32, self.mlp_f, self.mlp_b = e16, e32, mlp_f, mlp_b


WIMG_EDGE16, WIMG_EDGE32, WIMG_MLP_FWD, WIMG_MLP_BWD16 = 0, 1, 2, 3
USE_WEIGHT_IMAGES = True   # module attribute (tests/test_gpu_weight_images.py flips it): False = every launch stages its weights itself


@torch.no_grad()
def weight_images(blocks, grid3, basis, prec: str = "", with_backward: bool = True):
    """One launch for all pre-split weight images of a forward pass.  ``blocks``: [(conv_kernel_weight [64,64], n_dst of the block's edge
    convolution, (gamma, beta, W3, b3, W4, b4) of its ConvNeXt block)]; ``basis`` = (W1, b1, W2, b2) of the shared basis MLP.
    -> [WeightImages] in the order of ``blocks`` (None entries when switched off)."""
    import ctypes
    if not USE_WEIGHT_IMAGES or not blocks:
        return [None] * len(blocks)
    dev = grid3.device
    w1, b1, w2, b2 = [t.detach().contiguous() for t in basis]
    size = {k: hip.query("grl_wimg_bytes", k) for k in range(4)}
    jobs = []   # (kind, sources, block index, member)
    for i, (wk, n_dst, mlp) in enumerate(blocks):
        wk = wk.detach().contiguous()
        edge_src = [w1, b1, w2, b2, wk, grid3]
        fwd_kind = hip.query("grl_edge_fwd_image_kind", int(n_dst))
        if with_backward or fwd_kind == WIMG_EDGE16:
            jobs.append((WIMG_EDGE16, edge_src, i, "e16"))
        if fwd_kind == WIMG_EDGE32:
            jobs.append((WIMG_EDGE32, edge_src, i, "e32"))
        gamma, beta, w3, b3, w4, b4 = [t.detach().contiguous() for t in mlp]
        jobs.append((WIMG_MLP_FWD, [w3, b3, w4, b4, gamma, beta], i, "mlp_f"))
        if with_backward and w3.data_ptr() % 16 == 0:
            jobs.append((WIMG_MLP_BWD16, [w3, w4], i, "mlp_b"))
    offs, total = [], 0
    for kind, *_ in jobs:
        offs.append(total)
        total += (size[kind] + 255) & ~255
    buf = torch.empty(total, device=dev, dtype=torch.uint8)
    out = [WeightImages() for _ in blocks]
    views = []
    for (kind, srcs, i, member), o in zip(jobs, offs):
        v = buf[o:o + size[kind]]
        setattr(out[i], member, v)
        views.append(v)
    cap = hip.query("grl_wimg_max_jobs")
    if HEAD is not None and HEAD.wimg is None and 0 < len(jobs) <= cap:   # rides in the merged head launch (HeadLaunch)
        ptrs = []
        for kind, srcs, _, _ in jobs:
            ptrs += [t.data_ptr() for t in srcs] + [0] * (6 - len(srcs))
            HEAD.keep += list(srcs)
        HEAD.wimg = ([j[0] for j in jobs], ptrs, [v.data_ptr() for v in views])
        HEAD.keep.append(buf)
        return out
    for j0 in range(0, len(jobs), cap):
        part = jobs[j0:j0 + cap]
        n = len(part)
        ptrs = []
        for kind, srcs, _, _ in part:
            ptrs += [t.data_ptr() for t in srcs] + [0] * (6 - len(srcs))
        hip.call("grl_weight_images" + prec, n, (ctypes.c_int * n)(*[p_[0] for p_ in part]), (ctypes.c_void_p * (6 * n))(*ptrs),
                 (ctypes.c_void_p * n)(*[v.data_ptr() for v in views[j0:j0 + n]]))
    return out


class LiftEncode(torch.autograd.Function):
    """x[n,o,:] = [scalars | vectors . grid_o] W_enc^T   (reference hepi.py:136-143)."""

    @staticmethod
    def forward(ctx, scal, vec, grid3, w_enc, prec: str = ""):
        hip.check_f32(scal, vec, grid3, w_enc)
        n, s = scal.shape
        v = vec.shape[1]
        x = torch.empty(n, 16, 64, device=scal.device, dtype=hip.storage_dtype(prec))
        hip.call("grl_lift_encode_fwd" + prec, scal, vec, grid3, w_enc.contiguous(), x, n, s, v)
        ctx.save_for_backward(scal, vec, grid3)
        ctx.kf = s + v
        ctx.w_enc, ctx.prec = w_enc, prec
        return x

    @staticmethod
    def backward(ctx, dx):
        scal, vec, grid3 = ctx.saved_tensors
        n, s = scal.shape
        v = vec.shape[1]
        blocks = hip.query("grl_lift_bwd_blocks", n)
        partial = torch.empty(blocks, 64 * ctx.kf, device=dx.device, dtype=torch.float32)
        hip.check_latent(ctx.prec, dx)
        hip.call("grl_lift_encode_bwd" + ctx.prec, scal, vec, grid3, dx.contiguous(), partial, n, s, v)
        (dw,) = _emit_grads(partial, [(0, 64 * ctx.kf, (64, ctx.kf), ctx.w_enc)])
        return None, None, None, dw, None


class LiftEncodeMulti(torch.autograd.Function):
    """LiftEncode for every node type of a graph in ONE launch each way (the types share the encoder and the feature widths, reference
    hepi.py:136-143): apply(grid3, w_enc, prec, scal_0, vec_0, scal_1, vec_1, ...) -> (x_0, x_1, ...)."""

    @staticmethod
    def forward(ctx, grid3, w_enc, prec, *sv):
        import ctypes
        scal, vec = list(sv[0::2]), list(sv[1::2])
        hip.check_f32(grid3, w_enc, *scal, *vec)
        T = len(scal)
        s, v = scal[0].shape[1], vec[0].shape[1]
        if any(a.shape[1] != s or b.shape[1] != v for a, b in zip(scal, vec)):
            raise ValueError("LiftEncodeMulti: the node types must share the feature widths")
        dev = grid3.device
        xs = [torch.empty(a.shape[0], 16, 64, device=dev, dtype=hip.storage_dtype(prec)) for a in scal]
        ns = [int(a.shape[0]) for a in scal]
        hip.call("grl_lift_encode_fwd_multi" + prec, T, (ctypes.c_void_p * T)(*[a.data_ptr() for a in scal]),
                 (ctypes.c_void_p * T)(*[b.data_ptr() for b in vec]), grid3, w_enc.contiguous(),
                 (ctypes.c_void_p * T)(*[x.data_ptr() for x in xs]), (ctypes.c_int * T)(*ns), s, v)
        ctx.save_for_backward(grid3, *scal, *vec)
        ctx.kf, ctx.w_enc, ctx.prec, ctx.T, ctx.sv = s + v, w_enc, prec, T, (s, v)
        ctx.set_materialize_grads(False)   # a type whose latent never reaches the loss has no gradient: skipped, not zero-filled
        return tuple(xs)

    @staticmethod
    def backward(ctx, *dxs):
        import ctypes
        grid3, *rest = ctx.saved_tensors
        T = ctx.T
        scal, vec = rest[:T], rest[T:]
        s, v = ctx.sv
        dxs = [d.contiguous() if d is not None else None for d in dxs]
        for d in dxs:
            if d is not None:
                hip.check_latent(ctx.prec, d)
        ns = (ctypes.c_int * T)(*[int(a.shape[0]) if d is not None else 0 for a, d in zip(scal, dxs)])
        blocks = hip.query("grl_lift_bwd_blocks_multi", T, ns)
        if blocks == 0:
            return (None,) * (3 + 2 * T)
        partial = torch.empty(blocks, 64 * ctx.kf, device=grid3.device, dtype=torch.float32)
        if TAIL_PRE is not None and DEFERRED is not None and _all_leaf_grads([ctx.w_enc]):
            # (the fold of this slab is queued for the lane's tail anyway: the launch may wait for the fiber-basis backward and share its launch)
            _tail_pre_offer("lift", (ctx.prec, T, list(scal), list(vec), grid3, dxs, partial, ns, s, v))
        else:
            hip.call("grl_lift_encode_bwd_multi" + ctx.prec, T, (ctypes.c_void_p * T)(*[a.data_ptr() for a in scal]),
                     (ctypes.c_void_p * T)(*[b.data_ptr() for b in vec]), grid3,
                     (ctypes.c_void_p * T)(*[(d.data_ptr() if d is not None else 0) for d in dxs]), partial, ns, s, v)
        (dw,) = _emit_grads(partial, [(0, 64 * ctx.kf, (64, ctx.kf), ctx.w_enc)])
        return (None, dw, None) + (None,) * (2 * T)


# One-shot hook run right behind the NEXT edge convolution's forward launch (set by PolicyUpdater._plan_lanes, cleared when it fires).
AFTER_EDGE_HOOK = None


class EdgeConv(torch.autograd.Function):
    """x1[d] = sum_{e->d} Wk(basis_mlp(invariants_e)) * x_src[src(e)]   (reference hepi.py:145-157, conv.py:79-86,115-149)."""

    @staticmethod
    def forward(ctx, x_src, pos_src, pos_dst, grid3, w1, b1, w2, b2, wk, edges: EdgeSet, dim: int, residual=None, prec: str = "",
                wimg: Optional[WeightImages] = None):
        """``residual``: optional dict shared with the NodeMLP of the same layer when x_src is also that block's residual input:
        NodeMLP.backward leaves d(out)/d(x_dst) there and this backward adds it inside the d x_src kernel (no separate add pass).
        ``wimg``: the block's pre-split weight images of this pass (``weight_images``), reused by the backward."""
        hip.check_f32(pos_src, pos_dst, grid3, w1, b1, w2, b2, wk)
        hip.check_latent(prec, x_src)
        x1 = torch.empty(edges.n_dst, 16, 64, device=x_src.device, dtype=x_src.dtype)  # every row is written by the kernel
        args = [a.contiguous() for a in (w1, b1, w2, b2, wk)]
        sd = edges.split_d if SPLIT_FORWARD else None
        hip.call("grl_edge_conv_fwd_balanced" + prec, x_src, pos_src, pos_dst, edges.rowptr_d, edges.src_d, edges.dst_d, edges.n_dst, grid3,
                 dim, *args, x1, sd, (sd.numel() - 1) if sd is not None else 0, wimg.e16 if wimg else None, wimg.e32 if wimg else None,
                 rows=edges.n_edges * 16)
        global AFTER_EDGE_HOOK
        if AFTER_EDGE_HOOK is not None:   # behind the launch (PolicyUpdater: the signal the critic's lane starts on -- agent.py); the hook
            if AFTER_EDGE_HOOK():         # returns True once it has fired (it declines during the calibrating pass in front of the step's own)
                AFTER_EDGE_HOOK = None
        ctx.save_for_backward(x_src, pos_src, pos_dst, grid3, *args)
        ctx.edges, ctx.dim, ctx.residual, ctx.prec, ctx.wimg = edges, dim, residual, prec, wimg
        ctx.params = (w1, b1, w2, b2, wk)
        return x1

    @staticmethod
    def backward(ctx, dx1):
        x_src, pos_src, pos_dst, grid3, w1, b1, w2, b2, wk = ctx.saved_tensors
        e = ctx.edges
        dev = dx1.device
        blocks = hip.query("grl_edge_bwd_blocks", e.n_edges)
        psize = hip.query("grl_edge_partial_size")
        partial = torch.empty(blocks, psize, device=dev, dtype=torch.float32)
        dx_src = torch.empty_like(x_src)
        dres = ctx.residual.pop("dres", None) if ctx.residual is not None else None
        ss = e.split_s if SPLIT_BACKWARD else None
        hip.call("grl_edge_conv_bwd_balanced" + ctx.prec, x_src, pos_src, pos_dst, e.rowptr_d, e.src_d, e.dst_d, e.n_dst, e.n_edges, e.rowptr_s,
                 e.src_s, e.dst_s, e.n_src, grid3, ctx.dim, w1, b1, w2, b2, wk, dx1.contiguous(), dres, dx_src, partial,
                 ss, (ss.numel() - 1) if ss is not None else 0, ctx.wimg.e16 if ctx.wimg else None, rows=e.n_edges * 16)
        pw1, pb1, pw2, pb2, pwk = ctx.params
        dw1, db1, dw2, db2, dwk = _emit_grads(partial, [(0, 896, (64, 14), pw1), (896, 64, (64,), pb1), (960, 4096, (64, 64), pw2),
                                                         (5056, 64, (64,), pb2), (5120, 4096, (64, 64), pwk)])
        return (dx_src, None, None, None, dw1, db1, dw2, db2, dwk, None, None, None, None, None)


class EdgeMessages(torch.autograd.Function):
    """msg[e] = Wk(basis_mlp(invariants_e)) * x_src[src(e)] per edge, rows in destination-sorted edge order (conv.py:79,115-117) --
    the un-aggregated form FiberBundleConv(aggr="AttentionalAggregation") needs.  Backward: the fused d x_src / weight-gradient
    kernels of EdgeConv, fed with the per-edge gradient."""

    @staticmethod
    def forward(ctx, x_src, pos_src, pos_dst, grid3, w1, b1, w2, b2, wk, edges: EdgeSet, dim: int, residual=None, prec: str = ""):
        hip.check_f32(pos_src, pos_dst, grid3, w1, b1, w2, b2, wk)
        hip.check_latent(prec, x_src)
        msg = torch.empty(edges.n_edges, 16, 64, device=x_src.device, dtype=x_src.dtype)
        args = [a.contiguous() for a in (w1, b1, w2, b2, wk)]
        hip.call("grl_edge_messages_fwd" + prec, x_src, pos_src, pos_dst, edges.rowptr_d, edges.src_d, edges.dst_d, edges.n_dst,
                 edges.n_edges, grid3, dim, *args, msg, rows=edges.n_edges * 16)
        ctx.save_for_backward(x_src, pos_src, pos_dst, grid3, *args)
        ctx.edges, ctx.dim, ctx.residual, ctx.prec = edges, dim, residual, prec
        ctx.params = (w1, b1, w2, b2, wk)
        return msg

    @staticmethod
    def backward(ctx, dmsg):
        x_src, pos_src, pos_dst, grid3, w1, b1, w2, b2, wk = ctx.saved_tensors
        e = ctx.edges
        partial = torch.empty(hip.query("grl_edge_bwd_blocks", e.n_edges), hip.query("grl_edge_partial_size"), device=dmsg.device,
                              dtype=torch.float32)
        dx_src = torch.empty_like(x_src)
        dres = ctx.residual.pop("dres", None) if ctx.residual is not None else None
        hip.call("grl_edge_messages_bwd" + ctx.prec, x_src, pos_src, pos_dst, e.rowptr_d, e.src_d, e.dst_d, e.n_dst, e.n_edges,
                 e.rowptr_s, e.src_s, e.dst_s, e.s2d, e.n_src, grid3, ctx.dim, w1, b1, w2, b2, wk, dmsg.contiguous(), dres, dx_src,
                 partial, rows=e.n_edges * 16)
        pw1, pb1, pw2, pb2, pwk = ctx.params
        dw1, db1, dw2, db2, dwk = _emit_grads(partial, [(0, 896, (64, 14), pw1), (896, 64, (64,), pb1), (960, 4096, (64, 64), pw2),
                                                         (5056, 64, (64,), pb2), (5120, 4096, (64, 64), pwk)])
        return (dx_src, None, None, None, dw1, db1, dw2, db2, dwk, None, None, None, None)


class SoftmaxAggregate(torch.autograd.Function):
    """x1[d] = sum over the in-edges e of d of softmax_e(gate[e]) * msg[e], per orientation and channel (PyG AttentionalAggregation as
    conv.py:58-61,138-139 vmaps it; gate = gate_nn(msg) is computed by the caller)."""

    @staticmethod
    def forward(ctx, gate, msg, edges: EdgeSet, prec: str = ""):
        hip.check_f32(gate)
        hip.check_latent(prec, msg)
        x1 = torch.empty(edges.n_dst, 16, 64, device=msg.device, dtype=msg.dtype)
        gate = gate.contiguous()
        hip.call("grl_softmax_aggregate_fwd" + prec, gate, msg, edges.rowptr_d, edges.n_dst, x1)
        ctx.save_for_backward(gate, msg, x1)
        ctx.edges, ctx.prec = edges, prec
        return x1

    @staticmethod
    def backward(ctx, dx1):
        gate, msg, x1 = ctx.saved_tensors
        dgate, dmsg = torch.empty_like(gate), torch.empty_like(msg)
        hip.call("grl_softmax_aggregate_bwd" + ctx.prec, gate, msg, x1, dx1.contiguous(), ctx.edges.rowptr_d, ctx.edges.n_dst, dgate,
                 dmsg)
        return dgate, dmsg, None, None


class FiberConv(torch.autograd.Function):
    """x2[n,p,c] = 1/16 sum_o x1[n,o,c] fk[o,p,c] + bias[c]   (reference conv.py:88-90,108-109)."""

    @staticmethod
    def forward(ctx, x1, fk, bias, prec: str = ""):
        hip.check_f32(fk, bias)
        hip.check_latent(prec, x1)
        x2 = torch.empty_like(x1)
        fk = fk.contiguous()
        global PENDING_SIGNAL
        if PENDING_SIGNAL is not None and x1.shape[0] > 0:   # a lane signal rides on this launch (written when it starts: PolicyUpdater._plan_lanes)
            (fd, fs), PENDING_SIGNAL = PENDING_SIGNAL, None
            hip.call("grl_fiber_conv_fwd_sig" + prec, x1, fk, bias.contiguous(), x2, x1.shape[0], fd, fs)
        else:
            hip.call("grl_fiber_conv_fwd" + prec, x1, fk, bias.contiguous(), x2, x1.shape[0])
        global AFTER_FIBER_HOOK
        if AFTER_FIBER_HOOK is not None:   # experiment (bench.py --critic-gate fiber0): a one-shot hook behind the NEXT fiber convolution's launch
            if AFTER_FIBER_HOOK():
                AFTER_FIBER_HOOK = None
        ctx.save_for_backward(x1, fk)
        ctx.bias, ctx.prec = bias, prec
        return x2

    @staticmethod
    def backward(ctx, dx2):
        x1, fk = ctx.saved_tensors
        n = x1.shape[0]
        blocks = hip.query("grl_fiber_bwd_blocks", n)
        psize = hip.query("grl_fiber_partial_size")
        partial = torch.empty(blocks, psize, device=dx2.device, dtype=torch.float32)
        dx1 = torch.empty_like(x1)
        hip.check_latent(ctx.prec, dx2)
        hip.call("grl_fiber_conv_bwd" + ctx.prec, x1, fk, dx2.contiguous(), dx1, partial, n)
        dfk, dbias = _emit_grads(partial, [(0, 16 * 16 * 64, (16, 16, 64), None), (16 * 16 * 64, 64, (64,), ctx.bias)])
        return dx1, dfk, dbias, None


class FiberKernels(torch.autograd.Function):
    """fk_i = fiber_basis_fn(poly) @ Wf_i^T for all convolutions of a forward pass (reference hepi.py:109-123,157 / ponita.py:246-268,
    conv.py:62): parameter-only work on 256 rows, one launch forward, one launch + one reduction backward."""

    @staticmethod
    def forward(ctx, poly, w1, b1, w2, b2, *wfs):
        import ctypes
        hip.check_f32(poly, w1, b1, w2, b2, *wfs)
        n = len(wfs)
        dev = poly.device
        poly2 = poly.reshape(256, 3).contiguous()
        P = [t.contiguous() for t in (w1, b1, w2, b2)]
        W = [t.contiguous() for t in wfs]
        saved = torch.empty(4, 256, 64, device=dev, dtype=torch.float32)
        fks = [torch.empty(16, 16, 64, device=dev, dtype=torch.float32) for _ in range(n)]
        if HEAD is not None and HEAD.fiber is None:   # rides in the merged head launch (HeadLaunch)
            HEAD.fiber = (poly2, P, W, saved, fks)
        else:
            hip.call("grl_fiber_basis_fwd", poly2, *P, (ctypes.c_void_p * n)(*[t.data_ptr() for t in W]), n, saved,
                     (ctypes.c_void_p * n)(*[t.data_ptr() for t in fks]))
        ctx.save_for_backward(poly2, P[2], saved, *W)
        ctx.params = (w1, b1, w2, b2) + tuple(wfs)
        return tuple(fks)

    @staticmethod
    def backward(ctx, *dfks):
        import ctypes
        poly2, w2, saved, *W = ctx.saved_tensors
        n = len(W)
        dev = poly2.device
        d = [g.contiguous() if g is not None else None for g in dfks]
        partial = torch.empty(hip.query("grl_fiber_basis_blocks"), hip.query("grl_fiber_basis_partial_size", n), device=dev,
                              dtype=torch.float32)
        if TAIL_PRE is not None and DEFERRED is not None and _all_leaf_grads(ctx.params):
            _tail_pre_offer("fiber", (poly2, w2, list(W), saved, d, partial))
        else:
            hip.call("grl_fiber_basis_bwd", poly2, w2, (ctypes.c_void_p * n)(*[t.data_ptr() for t in W]), n, saved,
                     (ctypes.c_void_p * n)(*[(g.data_ptr() if g is not None else 0) for g in d]), partial)
        pw1, pb1, pw2, pb2, *pwf = ctx.params
        o = n * 4096
        segs = [(i * 4096, 4096, (64, 64), pwf[i]) for i in range(n)]
        segs += [(o, 4096, (64, 64), pw2), (o + 4096, 64, (64,), pb2), (o + 4160, 192, (64, 3), pw1), (o + 4352, 64, (64,), pb1)]
        g = _emit_grads(partial, segs)
        return (None, g[n + 2], g[n + 3], g[n], g[n + 1]) + tuple(g[:n])


def fiber_kernels(poly, basis_fn, convs):
    """{conv: fk} for the given FiberBundleConv-like modules (``.fiber_kernel.weight``); ``basis_fn`` = the reference's
    fiber_basis_fn Sequential (index 1 and 3 are the Linear layers), ``poly`` its (constant) polynomial input features."""
    out = {}
    for i in range(0, len(convs), 4):
        part = convs[i:i + 4]
        fks = FiberKernels.apply(poly, basis_fn[1].weight, basis_fn[1].bias, basis_fn[3].weight, basis_fn[3].bias,
                                 *[c.fiber_kernel.weight for c in part])
        out.update({id(c): fk for c, fk in zip(part, fks)})
    return out


class NodeMLP(torch.autograd.Function):
    """out = [prev +] x_dst + W4 GELU(W3 LN(x2) + b3) + b4   (reference conv.py:64-69,112; hetero_fiber_conv.py:63-64)."""

    @staticmethod
    def forward(ctx, x2, x_dst, gamma, beta, w3, b3, w4, b4, prev: Optional[torch.Tensor], residual=None, prec: str = "",
                wimg: Optional[WeightImages] = None):
        hip.check_f32(gamma, beta, w3, b3, w4, b4)
        hip.check_latent(prec, x2, x_dst, prev)
        ws = [a.contiguous() for a in (w3, b3, w4, b4, gamma, beta)]
        n_rows = x2.shape[0] * 16
        out = prev.clone() if prev is not None else torch.empty_like(x2)
        hip.call("grl_node_mlp_fwd_img" + prec, x2, x_dst, *ws, out, n_rows, 1 if prev is not None else 0, wimg.mlp_f if wimg else None,
                 rows=n_rows)
        ctx.save_for_backward(x2, *ws)
        ctx.has_prev, ctx.prec, ctx.wimg = prev is not None, prec, wimg
        ctx.residual = residual
        ctx.params = (w3, b3, w4, b4, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, w3, b3, w4, b4, gamma, beta = ctx.saved_tensors
        dev = dout.device
        dout = dout.contiguous()
        n_rows = x2.shape[0] * 16
        dx2 = torch.empty_like(x2)
        blocks = hip.query("grl_node_mlp_bwd_blocks", n_rows)
        psize = hip.query("grl_node_mlp_partial_size")
        partial = torch.empty(blocks + 1, psize, device=dev, dtype=torch.float32)   # last row: scratch (shared W3 fragment image)
        hip.call("grl_node_mlp_bwd_img" + ctx.prec, x2, dout, w3, b3, w4, b4, gamma, beta, dx2, partial, n_rows,
                 ctx.wimg.mlp_b if ctx.wimg else None, rows=n_rows)
        partial = partial[:blocks]
        pw3, pb3, pw4, pb4, pg, pbt = ctx.params
        dw3, db3, dw4, db4, dgam, dbet = _emit_grads(partial, [(0, 16384, (256, 64), pw3), (16384, 256, (256,), pb3),
                                                               (16640, 16384, (64, 256), pw4), (33024, 64, (64,), pb4),
                                                               (33088, 64, (64,), pg), (33152, 64, (64,), pbt)])
        d_dst = dout
        if ctx.residual is not None:   # handed to the EdgeConv backward of the same layer (same tensor x feeds both)
            ctx.residual["dres"] = dout
            d_dst = None
        return (dx2, d_dst, dgam, dbet, dw3, db3, dw4, db4, dout if ctx.has_prev else None, None, None, None)


class Readout(torch.autograd.Function):
    """Decoder + orientation pooling + contextual std head on the actuator nodes
    (reference hepi.py:173-190, gnn_gaussian_policy_diag.py:65-87) -> (mean [N, ov, 3], sigma [N, 3 ov], hidden [N, 64])."""

    @staticmethod
    def forward(ctx, lat, grid3, wd, bd, ws, bs, shift: float, min_std: float, od: int, ov: int):
        hip.check_f32(lat, grid3, wd, bd, ws, bs)   # (bf16 latents of the few actuator nodes are widened by the caller)
        n = lat.shape[0]
        dev = lat.device
        mean = torch.empty(n, ov, 3, device=dev, dtype=torch.float32)
        sigma = torch.empty(n, 3 * ov, device=dev, dtype=torch.float32)
        hidden = torch.empty(n, 64, device=dev, dtype=torch.float32)
        ws_, bs_ = ws.contiguous(), bs.contiguous()
        wd_, bd_ = wd.contiguous(), bd.contiguous()
        hip.call("grl_readout_fwd", lat, grid3, wd_, bd_, ws_, bs_, float(shift), float(min_std), mean, sigma, hidden, n, od, ov)
        ctx.save_for_backward(lat, grid3, wd_, bd_, ws_, bs_)
        ctx.cfg = (float(shift), od, ov)
        ctx.params = (wd, bd, ws, bs)
        ctx.set_materialize_grads(False)   # ``hidden`` is rarely used: no zero tensor (a fill launch on the step's chain) for its gradient
        return mean, sigma, hidden

    @staticmethod
    def backward(ctx, dmean, dsigma, dhidden):
        lat, grid3, wd, bd, ws, bs = ctx.saved_tensors
        shift, od, ov = ctx.cfg
        n = lat.shape[0]
        dev = lat.device
        blocks = hip.query("grl_readout_blocks", n)
        psize = hip.query("grl_readout_partial_size")
        partial = torch.empty(blocks, psize, device=dev, dtype=torch.float32)
        dlat = torch.empty_like(lat)
        dmean = torch.zeros(n, ov, 3, device=dev) if dmean is None else dmean.contiguous()
        dsigma = torch.zeros(n, 3 * ov, device=dev) if dsigma is None else dsigma.contiguous()
        dh = dhidden.contiguous() if dhidden is not None else None
        hip.call("grl_readout_bwd", lat, grid3, wd, bd, ws, bs, shift, dmean, dsigma, dh, dlat, partial, n, od, ov)
        J, aper = od + ov, 3 * ov
        pwd, pbd, pws, pbs = ctx.params
        dwd, dbd, dws, dbs = _emit_grads(partial, [(0, J * 64, (J, 64), pwd), (256, J, (J,), pbd), (260, aper * 64, (aper, 64), pws),
                                                    (260 + 384, aper, (aper,), pbs)])
        return dlat, None, dwd, dbd, dws, dbs, None, None, None, None


class DeepSetsPipeline:
    """The six kernels of the DeepSets critic (reference deepsets.py:34-53, gnn_vf_net.py:50-86) as explicit stages.  Between
    fwd1/fwd2, fwd2/fwd3, bwd3/bwd2 and bwd2/bwd1 the whole-tensor LayerNorm sums (``stats`` / ``bst`` halves) have to be summed
    over the data-parallel ranks by the caller; ``world`` only enters through the element counts."""

    PARAM_ORDER = ("w1", "b1", "g1", "be1", "w2", "b2", "w3", "b3", "g2", "be2", "w4", "b4", "wv", "bv")

    def __init__(self, x, params, world=1):
        hip.check_f32(x, *params)
        self.B, self.n, self.d = x.shape
        B, n, dev = self.B, self.n, x.device
        self.x = x.contiguous()
        self.P = [t.contiguous() for t in params]
        # per-workgroup (sum, sum of squares) slots, written in full by the producing stage: no zeroing, no atomics
        ns = 2 * hip.query("grl_deepsets_stat_slots")
        slots = torch.empty(4 * ns, device=dev, dtype=torch.float64)
        self.stats1, self.stats2 = slots[0:ns], slots[ns:2 * ns]          # (h1, h1^2), (u1, u1^2)
        self.bst2, self.bst1 = slots[2 * ns:3 * ns], slots[3 * ns:4 * ns]  # (q2, q2 xh2), (q1, q1 xh1)
        self.h1 = torch.empty(B, n, 64, device=dev, dtype=torch.float32)
        self.z = torch.empty(B, 64, device=dev, dtype=torch.float32)
        self.u1 = torch.empty(B, 64, device=dev, dtype=torch.float32)
        self.value = torch.empty(B, device=dev, dtype=torch.float32)
        self.c1, self.c2 = float(B * world * n * 64), float(B * world * 64)

    def fwd1(self):
        w1, b1 = self.P[0], self.P[1]
        hip.call("grl_deepsets_fwd1", self.x, w1, b1, self.h1, self.stats1, self.B, self.n, self.d)

    def fwd2(self):
        w1, b1, g1, be1, w2, b2, w3, b3 = self.P[:8]
        hip.call("grl_deepsets_fwd2", self.h1, self.stats1, ctypes_double(self.c1), g1, be1, w2, b2, w3, b3, self.z, self.u1,
                 self.stats2, self.B, self.n)

    def fwd3(self):
        g2, be2, w4, b4, wv, bv = self.P[8:]
        hip.call("grl_deepsets_fwd3", self.u1, self.stats2, ctypes_double(self.c2), g2, be2, w4, b4, wv, bv, self.value, self.B)
        return self.value

    def bwd3(self, dvalue):
        dev = self.x.device
        blocks = hip.query("grl_deepsets_blocks", self.B)
        self.part3 = torch.empty(blocks, hip.query("grl_deepsets_partial3"), device=dev)
        self.part2 = torch.empty(blocks, hip.query("grl_deepsets_partial2"), device=dev)
        self.part1 = torch.empty(blocks, 64 * self.d + 64, device=dev)
        self.q2 = torch.empty(self.B, 64, device=dev)
        self.q1 = torch.empty(self.B, self.n, 64, device=dev)
        g2, be2, w4, b4, wv, bv = self.P[8:]
        hip.call("grl_deepsets_bwd3", self.u1, self.stats2, ctypes_double(self.c2), g2, be2, w4, b4, wv, dvalue.contiguous(),
                 self.q2, self.bst2, self.part3, self.B)

    def bwd2(self):
        w1, b1, g1, be1, w2, b2, w3, b3 = self.P[:8]
        hip.call("grl_deepsets_bwd2", self.h1, self.stats1, ctypes_double(self.c1), g1, be1, w2, w3, self.z, self.u1,
                 self.stats2, ctypes_double(self.c2), self.q2, self.bst2, self.q1, self.bst1, self.part2, self.B, self.n)

    def bwd1(self, leaves):
        """Last stage + folding of the partial slabs.  ``leaves``: the 14 parameter tensors in PARAM_ORDER (gradients are
        accumulated in place into ``.grad`` where that buffer exists, see _emit_grads).  Returns the 14 gradients (or None)."""
        d = self.d
        hip.call("grl_deepsets_bwd1", self.x, self.h1, self.stats1, ctypes_double(self.c1), self.q1, self.bst1, self.part1,
                 self.B, self.n, d)
        (pw1, pb1, pg1, pbe1, pw2, pb2, pw3, pb3, pg2, pbe2, pw4, pb4, pwv, pbv) = leaves
        dw4, db4, dwv, dbv, dg2, dbe2 = _emit_grads(self.part3, [(0, 4096, (64, 64), pw4), (4096, 64, (64,), pb4),
                                                                 (4160, 64, (1, 64), pwv), (4352, 1, (1,), pbv),
                                                                 (4224, 64, (64,), pg2), (4288, 64, (64,), pbe2)])
        dw3, db3, dw2, db2, dg1, dbe1 = _emit_grads(self.part2, [(0, 4096, (64, 64), pw3), (4096, 64, (64,), pb3),
                                                                 (4160, 4096, (64, 64), pw2), (8256, 64, (64,), pb2),
                                                                 (8320, 64, (64,), pg1), (8384, 64, (64,), pbe1)])
        dw1, db1 = _emit_grads(self.part1, [(0, 64 * d, (64, d), pw1), (64 * d, 64, (64,), pb1)])
        return (dw1, db1, dg1, dbe1, dw2, db2, dw3, db3, dg2, dbe2, dw4, db4, dwv, dbv)


@torch.no_grad()
def deepsets_values_groups(x: torch.Tensor, params, groups: int, group=None) -> torch.Tensor:
    """No-grad critic pass over ``groups`` independent batches in three launches: x [groups * B, n, d] (group-major) -> V [groups, B].
    Every group has its own whole-tensor LayerNorm statistics, exactly as if ``DeepSetsValue`` had been called once per group
    (gnn_vf_net.py:72-80 loops over the time steps of a [N, T, .] input); the per-group launch grid is the single-batch grid, so the
    values are bitwise those of the loop.  ``params``: the 14 tensors in DeepSetsPipeline.PARAM_ORDER.
    ``group``: torch.distributed process group of a data-parallel run -- every rank holds B of the world * B samples of each group; the
    [groups, slots] statistic arrays are summed over the ranks by ONE all-reduce per LayerNorm stage (two per call, whatever ``groups``)."""
    hip.check_f32(x, *params)
    GB, n, d = x.shape
    assert GB % groups == 0
    B, dev = GB // groups, x.device
    world = 1
    if group is not None:
        import torch.distributed as dist
        world = dist.get_world_size(group)
    P = [t.contiguous() for t in params]
    ns = 2 * hip.query("grl_deepsets_stat_slots")
    slots = torch.empty(2, groups, ns, device=dev, dtype=torch.float64)
    h1 = torch.empty(GB, n, 64, device=dev, dtype=torch.float32)
    z = torch.empty(GB, 64, device=dev, dtype=torch.float32)
    u1 = torch.empty(GB, 64, device=dev, dtype=torch.float32)
    value = torch.empty(groups, B, device=dev, dtype=torch.float32)
    w1, b1, g1, be1, w2, b2, w3, b3, g2, be2, w4, b4, wv, bv = P
    hip.call("grl_deepsets_fwd1_groups", x.contiguous(), w1, b1, h1, slots[0], B, n, d, groups)
    if world > 1:
        dist.all_reduce(slots[0], group=group)
    hip.call("grl_deepsets_fwd2_groups", h1, slots[0], ctypes_double(float(B * world * n * 64)), g1, be1, w2, b2, w3, b3, z, u1, slots[1], B, n,
             groups)
    if world > 1:
        dist.all_reduce(slots[1], group=group)
    hip.call("grl_deepsets_fwd3_groups", u1, slots[1], ctypes_double(float(B * world * 64)), g2, be2, w4, b4, wv, bv, value, B, groups)
    return value


class DeepSetsValue(torch.autograd.Function):
    """DeepSets critic + value head as one autograd node: x [B, n, d] -> V [B] (stages: DeepSetsPipeline).

    ``group``: optional torch.distributed process group; the whole-tensor LayerNorm statistics (and their backward
    counterparts) are all-reduced over it so a sharded minibatch reproduces the single-device result."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, w2, b2, w3, b3, g2, be2, w4, b4, wv, bv, group):
        leaves = (w1, b1, g1, be1, w2, b2, w3, b3, g2, be2, w4, b4, wv, bv)
        world = 1
        if group is not None:
            import torch.distributed as dist
            world = dist.get_world_size(group)
        pipe = DeepSetsPipeline(x, leaves, world)
        pipe.fwd1()
        if world > 1:
            dist.all_reduce(pipe.stats1, group=group)
        pipe.fwd2()
        if world > 1:
            dist.all_reduce(pipe.stats2, group=group)
        value = pipe.fwd3()
        ctx.pipe, ctx.leaves, ctx.group, ctx.world = pipe, leaves, group, world
        return value

    @staticmethod
    def backward(ctx, dvalue):
        pipe, group, world = ctx.pipe, ctx.group, ctx.world
        pipe.bwd3(dvalue)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(pipe.bst2, group=group)
        pipe.bwd2()
        if world > 1:
            dist.all_reduce(pipe.bst1, group=group)
        grads = pipe.bwd1(ctx.leaves)
        return (None,) + tuple(grads) + (None,)


def ctypes_double(v: float):
    import ctypes
    return ctypes.c_double(v)


TRPL_SUM_KEYS = ("loss_objective", "loss_trust_region", "entropy_dist", "loss_critic", "sum_w", "sum_w2", "mean_constraint",
                 "cov_constraint", "entropy", "entropy_diff", "count", "kl")


def trpl_fwd_bwd(loc, sigma, batch, value, *, mean_bound, cov_bound, trust_region_coeff, entropy_coef, critic_coef, clip_value,
                 global_batch: int, adv_stats: Optional[torch.Tensor], want_projection: bool = False, sums=None, maxes=None,
                 proj_type: int = 0, defer_fold: bool = False, adv_local: bool = False):
    """Launches the fused TRPL kernel (proj_type 0 KL | 1 Frobenius | 2 Wasserstein).  Returns (sums fp64[12], maxes u32[2], dloc,
    dsigma, dvalue, proj_mean, proj_var).  ``defer_fold``: the per-workgroup slots are not folded into ``sums`` / ``maxes`` by this call;
    the returned ``sums`` is then a callable that does it (on whatever stream is current when it is called) and returns (sums, maxes)."""
    import ctypes
    hip.check_f32(loc, sigma)
    B, A = loc.shape
    dev = loc.device
    cfg = (ctypes.c_double * 10)(mean_bound, cov_bound, trust_region_coeff, entropy_coef, critic_coef,
                                 clip_value if clip_value else 0.0, 1.0 / global_batch, float(global_batch), float(proj_type),
                                 1.0 if adv_local else 0.0)   # adv_local: the batch's advantage statistics are summed inside the kernel
    if sums is None:   # otherwise: views of the caller's per-step workspace (written in full by the launch)
        sums = torch.empty(12, device=dev, dtype=torch.float64)
        maxes = torch.empty(2, device=dev, dtype=torch.int32)
    slots = torch.empty(hip.query("grl_trpl_slot_doubles", B), device=dev, dtype=torch.float64)   # per-workgroup sums
    dloc, dsigma = torch.empty_like(loc), torch.empty_like(sigma)
    dvalue = torch.empty(B, device=dev, dtype=torch.float32) if value is not None else None
    pm = torch.empty_like(loc) if want_projection else None
    pv = torch.empty_like(loc) if want_projection else None
    f = lambda t: t.reshape(B, -1).contiguous() if t.dim() > 1 else t.contiguous()
    hip.call("grl_trpl_fwd_bwd", cfg, A, loc.contiguous(), sigma.contiguous(), f(batch["action"]), f(batch["loc"]), f(batch["var"]),
             batch["sample_log_prob"].reshape(B).contiguous(), batch["advantage"].reshape(B).contiguous(),
             value.reshape(B).contiguous() if value is not None else None,
             batch["state_value"].reshape(B).contiguous() if value is not None else None,
             batch["value_target"].reshape(B).contiguous() if value is not None else None,
             dloc, dsigma, dvalue, pm, pv, adv_stats, None if defer_fold else sums, maxes, slots, B)
    if defer_fold:
        def fold(sums=sums, maxes=maxes, slots=slots):
            hip.call("grl_trpl_fold", slots, B, sums, maxes)
            return sums, maxes
        fold.slots, fold.batch, fold.sums, fold.maxes = slots, B, sums, maxes   # (for a caller that folds and reports in one launch)
        return fold, maxes, dloc, dsigma, dvalue, pm, pv
    return sums, maxes, dloc, dsigma, dvalue, pm, pv


def trpl_target_terms(loc, sigma, tgt_mean, tgt_S, *, mean_bound, cov_bound, trust_region_coeff, global_batch: int, proj_type: int = 0):
    """Trust-region measure of (p, detached target) with its gradient -- the fused kernel with the projection skipped
    (grl_trpl_target_terms).  ``sigma`` = sqrt of the policy's covariance diagonal, ``tgt_S`` = the target's covariance diagonal.
    Returns (sums fp64[12], maxes u32[2], dloc, dsigma)."""
    import ctypes
    hip.check_f32(loc, sigma, tgt_mean, tgt_S)
    B, A = loc.shape
    dev = loc.device
    cfg = (ctypes.c_double * 10)(mean_bound, cov_bound, trust_region_coeff, 0.0, 0.0, 0.0, 1.0 / global_batch, float(global_batch),
                                 float(proj_type), 0.0)
    sums = torch.empty(12, device=dev, dtype=torch.float64)
    maxes = torch.empty(2, device=dev, dtype=torch.int32)
    slots = torch.empty(hip.query("grl_trpl_slot_doubles", B), device=dev, dtype=torch.float64)
    dloc, dsigma = torch.empty_like(loc), torch.empty_like(sigma)
    zeros_b = torch.zeros(B, device=dev, dtype=torch.float32)
    hip.call("grl_trpl_target_terms", cfg, A, loc.contiguous(), sigma.contiguous(), tgt_mean.contiguous(), tgt_S.contiguous(), dloc,
             dsigma, sums, maxes, slots, zeros_b, B)
    return sums, maxes, dloc, dsigma
