"""Collector-side running normalisation + clip against the CPU restatement, over successive calls (the decayed
statistics carry over) and with frozen statistics."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_observation_normalizer_matches_oracle():
    from oracle import transforms as otf
    from geometry_rl_amd.transforms import ObservationNormalizer
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    norm = ObservationNormalizer(device=dev)
    st = {"position_vectors": otf.VecNormState(3), "velocity_vectors": otf.VecNormState(3), "scalars": otf.VecNormState(5)}
    for call in range(4):
        B = 257 + 31 * call   # ragged batch sizes between calls
        obs = {"position_vectors": torch.randn(B, 3 * 65, generator=g) * (3 + call) + 0.5,
               "velocity_vectors": torch.randn(B, 12, generator=g) * 30,
               "scalars": torch.randn(B, 5, generator=g) * 2 - 1}
        update = call < 3
        out = norm({k: v.to(dev) for k, v in obs.items()}, update=update)
        for k in ("position_vectors", "velocity_vectors"):
            ref_n = otf.clip(otf.vecnorm_update(obs[k].reshape(B, -1, 3), st[k], 0.99999, 1e-2, update).reshape(B, -1), -20.0, 20.0)
            assert (out["norm_" + k].cpu() - ref_n).abs().max().item() <= 2e-5, (call, k)
            assert torch.equal(out[k].cpu(), otf.clip(obs[k], -20.0, 20.0))
        ref_s = otf.clip(otf.vecnorm_update(obs["scalars"], st["scalars"], 0.99999, 1e-2, update), -20.0, 20.0)
        assert (out["scalars"].cpu() - ref_s).abs().max().item() <= 2e-5, call
    s = norm.state["position_vectors"].cpu()
    ref = st["position_vectors"]
    assert torch.allclose(s[:3], ref.sum, rtol=1e-5) and torch.allclose(s[3:6], ref.ssq, rtol=1e-5) and torch.allclose(s[6:], ref.count)
