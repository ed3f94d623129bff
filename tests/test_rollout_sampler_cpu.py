"""Host logic of the rollout driver (no GPU): the env-aligned sampler draws every frame exactly once per epoch."""
import torch


def test_epoch_indices_cover_every_frame_once():
    from geometry_rl_amd.rollout import RolloutDriver
    drv = RolloutDriver(updater=None, spec=None, ppo_epochs=2, seed=3)
    N, T = 7, 5
    seen = []
    for _ in range(2):
        idx = drv.epoch_indices(N, T, torch.device("cpu"))
        assert idx.shape == (T, N) and idx.dtype == torch.int64
        assert sorted(idx.reshape(-1).tolist()) == list(range(N * T))       # without replacement, whole rollout
        assert torch.equal(idx // T, torch.arange(N)[None, :].expand(T, N))  # position n of every minibatch = environment n
        seen.append(idx)
    assert not torch.equal(seen[0], seen[1])  # a fresh permutation per epoch
