"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the collector-side observation transforms.

Follows geometry_rl/torchrl/envs/transforms.py:141-163 (``NDVecNorm._call``: statistics of the shape given by ``shapes``, batch and
point dimensions counted into N by ``_count_left`` :63-69) on top of torchrl 0.3.1 ``VecNorm._update`` [upstream, absent from
/root/reference: restated from its published behaviour -- parity unpinned], then ``ClipTransform`` as configured in
configs/rigid_insertion_multi_hepi_trpl_cfg.yaml:47-72 (decay 0.99999, eps 1e-2, clip +-20)."""
import torch


class VecNormState:
    def __init__(self, K: int):
        self.sum = torch.zeros(K)
        self.ssq = torch.zeros(K)
        self.count = torch.zeros(1)


def vecnorm_update(x: torch.Tensor, st: VecNormState, decay: float, eps: float, update: bool = True) -> torch.Tensor:
    """x [..., K]; statistics over every leading dimension (VecNorm._update: sum/ssq/count with exponential decay)."""
    K = st.sum.numel()
    v = x.reshape(-1, K)
    if update:
        st.sum = st.sum * decay + v.sum(0)
        st.ssq = st.ssq * decay + v.pow(2).sum(0)
        st.count = st.count * decay + v.shape[0]
    mean = st.sum / st.count
    std = (st.ssq / st.count - mean.pow(2)).clamp_min(eps).sqrt()
    return ((v - mean) / std.clamp_min(eps)).reshape(x.shape)


def clip(x: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    return x.clamp(lo, hi)
