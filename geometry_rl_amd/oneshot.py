"""One-shot all-reduce over directly mapped peer buffers (csrc/oneshot.hip, include/grl_hip.h ``grl_oneshot_allreduce``) -- the
latency-bound alternative to RCCL's ring for the data-parallel step's one bandwidth-relevant collective (the
reference has no distributed code, examples/torchrl/train.py:304-316 only fixes what every replica must end up with).

Two ways to obtain the mapped areas:

* ``local_ranks(world, n, device)``: W stand-in "ranks" inside ONE process on one GPU (each with its own payload, staging rows and flags,
  driven from W streams) -- how the protocol is tested without a multi-GPU node (tests/test_gpu_oneshot.py);
* ``ipc_rank(group, n, device)``: one process per GPU; every rank allocates its area, the ranks exchange torch's IPC handles
  (``UntypedStorage._share_cuda_`` = hipIpcGetMemHandle; needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this stack) over the process group and map
  each other's areas.  Used by ``PolicyUpdater`` only with GRL_DP_ONESHOT=1 -- OFF by default: it has never run on more than one GPU
  (DESIGN.md section 5 holds the switch-on criterion).
"""
import ctypes
from typing import List

import torch

from . import hip


class OneShotRank:
    """One rank's view: the areas of all ranks as mapped here (``payloads`` / ``stages`` / ``flags``: lists of ``world`` tensors), its own
    index, the call counter.  ``payload`` (= payloads[rank]) is the tensor that is reduced IN PLACE."""

    def __init__(self, payloads: List[torch.Tensor], stages: List[torch.Tensor], flags: List[torch.Tensor], rank: int, timeout_ms: int = 2000):
        self.world, self.rank = len(payloads), rank
        self.payloads, self.stages, self.flags = payloads, stages, flags
        self.payload = payloads[rank]
        self.n = int(self.payload.numel())
        if self.n % 4 or any(int(p.numel()) != self.n for p in payloads):
            raise ValueError("one-shot all-reduce: payloads of equal length, a multiple of 4 floats")
        if any(int(s.numel()) < hip.query("grl_oneshot_stage_floats", self.n, self.world) for s in stages):
            raise ValueError("staging area too small (grl_oneshot_stage_floats)")
        self.status = torch.zeros(1, device=self.payload.device, dtype=torch.int32)
        self.seq, self.timeout_ms = 0, int(timeout_ms)
        W = self.world
        self._ptrs = ((ctypes.c_void_p * W)(*[t.data_ptr() for t in payloads]), (ctypes.c_void_p * W)(*[t.data_ptr() for t in stages]),
                      (ctypes.c_void_p * W)(*[t.data_ptr() for t in flags]))

    def all_reduce(self):
        """Enqueue this rank's side on the current stream (every rank must; the kernels meet on the device)."""
        self.seq += 1
        hip.call("grl_oneshot_allreduce", *self._ptrs, self.rank, self.world, self.n, ctypes.c_uint(self.seq), self.timeout_ms, self.status)

    def check(self):
        """Synchronises: raises if a wait of any call since the last check ran into its timeout."""
        code = int(self.status.item())
        if code:
            self.status.zero_()
            raise RuntimeError(f"one-shot all-reduce timed out on rank {self.rank} ({'a contribution' if code == 1 else 'a result chunk'} did not "
                               f"arrive within {self.timeout_ms} ms): not every rank enqueued call {self.seq}, or the kernels were not resident together")


def all_reduce_local(ranks: List["OneShotRank"]):
    """All stand-in ranks of ``local_ranks`` in ONE launch (grl_oneshot_allreduce_local): their workgroups are resident together by
    construction.  (W separate launches on W streams of one process may land on one hardware queue and serialise -- then every wait runs
    into its timeout: that form is only safe across processes / devices.)"""
    r0 = ranks[0]
    seq = max(r.seq for r in ranks) + 1
    for r in ranks:
        r.seq = seq
    status = torch.zeros(len(ranks), device=r0.payload.device, dtype=torch.int32)
    hip.call("grl_oneshot_allreduce_local", *r0._ptrs, r0.world, r0.n, ctypes.c_uint(seq), r0.timeout_ms, status)
    return status


def _areas(n, world, device):
    stage = torch.empty(hip.query("grl_oneshot_stage_floats", n, world), device=device, dtype=torch.float32)
    flags = torch.zeros(hip.query("grl_oneshot_flag_words", world), device=device, dtype=torch.int32)
    return stage, flags


def local_ranks(world: int, n: int, device, timeout_ms: int = 2000) -> List[OneShotRank]:
    """W stand-in ranks on ONE device (single process): -> [OneShotRank]; fill ``r.payload`` and call ``r.all_reduce()`` from W streams."""
    payloads = [torch.zeros(n, device=device, dtype=torch.float32) for _ in range(world)]
    st_fl = [_areas(n, world, device) for _ in range(world)]
    return [OneShotRank(payloads, [s for s, _ in st_fl], [f for _, f in st_fl], r, timeout_ms) for r in range(world)]


def ipc_rank(group, payload: torch.Tensor, timeout_ms: int = 2000) -> OneShotRank:
    """One process per GPU: map every rank's (payload | staging | flags) through torch's CUDA IPC and return this rank's view.  ``payload``
    must be a WHOLE allocation of its own (offset 0 of its storage), float32, a multiple of 4 elements.  Collective over ``group``."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = int(payload.numel())
    if payload.storage_offset() != 0 or payload.dtype != torch.float32 or not payload.is_contiguous():
        raise ValueError("ipc_rank: the payload must start its own allocation (a fresh torch.zeros / torch.empty tensor)")
    stage, flags = _areas(n, world, payload.device)
    torch.cuda.synchronize(payload.device)
    mine = [t.untyped_storage()._share_cuda_() for t in (payload, stage, flags)]
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    payloads, stages, flag_ts = [], [], []
    for r in range(world):
        if r == rank:
            payloads.append(payload); stages.append(stage); flag_ts.append(flags)
            continue
        ts = []
        for handle, like in zip(everyone[r], (payload, stage, flags)):
            st = torch.UntypedStorage._new_shared_cuda(*handle)
            ts.append(torch.empty(0, dtype=like.dtype, device=st.device).set_(st, 0, like.shape))
        payloads.append(ts[0]); stages.append(ts[1]); flag_ts.append(ts[2])
    dist.barrier(group=group)   # nobody raises a flag before everybody has zeroed and mapped
    return OneShotRank(payloads, stages, flag_ts, rank, timeout_ms)
