"""Collector-side observation transforms on the device: running normalisation + clipping of raw rollouts, so
pre-recorded *raw* observations can be replayed through the same input distribution the reference trains on.

Mirror of the transform stack of configs/rigid_insertion_multi_hepi_trpl_cfg.yaml:47-72:
``ReshapeTransform([-1, 3])`` + ``NDVecNorm(in=[position_vectors, velocity_vectors], out=[norm_*], shapes=[3, 3])`` (one statistic
per axis, shared by all points and environments: geometry_rl/torchrl/envs/transforms.py:63-69,141-163) + ``VecNorm(scalars)`` +
``FlattenObservation`` + ``ClipTransform(low, high)`` on raw and normalised groups.  One fused call per group
(``grl_vecnorm``: column statistics, decayed state update, standardise, clip)."""
import ctypes
from typing import Dict, Iterable

import torch

from . import hip


class ObservationNormalizer:
    def __init__(self, vector_keys: Iterable[str] = ("position_vectors", "velocity_vectors"), scalar_keys: Iterable[str] = ("scalars",),
                 decay: float = 0.99999, eps: float = 1e-2, low: float = -20.0, high: float = 20.0, device="cuda"):
        self.vector_keys, self.scalar_keys = list(vector_keys), list(scalar_keys)
        self.decay, self.eps, self.low, self.high = float(decay), float(eps), float(low), float(high)
        self.device = torch.device(device)
        self.state: Dict[str, torch.Tensor] = {}   # key -> float32 [2K+1] = [sum | ssq | count]
        self._scratch: Dict[int, torch.Tensor] = {}

    def _run(self, key: str, x: torch.Tensor, K: int, update: bool, want_norm: bool, want_clip: bool):
        hip.check_f32(x)
        x = x.contiguous()
        if key not in self.state:
            self.state[key] = torch.zeros(2 * K + 1, device=x.device, dtype=torch.float32)
        if K not in self._scratch:
            self._scratch[K] = torch.empty(hip.query("grl_vecnorm_scratch_bytes", K), device=x.device, dtype=torch.uint8)
        y_norm = torch.empty_like(x) if want_norm else None
        y_clip = torch.empty_like(x) if want_clip else None
        hip.call("grl_vecnorm", x, ctypes.c_longlong(x.numel() // K), K, self.decay, self.eps, bool(update), self.low, self.high,
                 self.state[key], self._scratch[K], y_norm, y_clip)
        return y_norm, y_clip

    def __call__(self, obs: Dict[str, torch.Tensor], update: bool = True) -> Dict[str, torch.Tensor]:
        """obs groups [B, D] (flat, as the loss consumes them) -> same dict with ``norm_<vector key>`` added, scalars normalised in
        place of the raw values (VecNorm without out_keys) and every group clipped."""
        out = dict(obs)
        for k in self.vector_keys:
            yn, yc = self._run(k, obs[k], 3, update, True, True)
            out["norm_" + k], out[k] = yn, yc
        for k in self.scalar_keys:
            yn, _ = self._run(k, obs[k], obs[k].shape[-1], update, True, False)
            out[k] = yn
        return out
