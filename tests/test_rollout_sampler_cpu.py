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


def test_minibatch_size_multiples_stay_env_aligned():
    """mini_batch_size = 2 * num_envs (configs/cloth_hanging_multi_hepi_trpl_cfg.yaml:40,125): T/2 minibatches per epoch, every frame
    once, and row i of every minibatch belongs to environment i mod N (the invariant of the topology cached per batch size)."""
    from types import SimpleNamespace
    from geometry_rl_amd.rollout import RolloutDriver
    N, T = 6, 8
    drv = RolloutDriver(updater=None, spec=SimpleNamespace(family="cloth"), seed=1, mini_batch_size=2 * N)
    mbs = drv.epoch_minibatches(N, T, torch.device("cpu"))
    assert len(mbs) == T // 2 and all(m.numel() == 2 * N for m in mbs)
    assert sorted(torch.cat(mbs).tolist()) == list(range(N * T))
    for m in mbs:
        assert torch.equal(m // T, torch.arange(N).repeat(2))
    import pytest
    with pytest.raises(ValueError):
        RolloutDriver(updater=None, spec=None, mini_batch_size=N + 1).epoch_minibatches(N, T, torch.device("cpu"))


def test_uniform_sampling_is_the_reference_sampler_and_guarded_for_ragged_tasks():
    from types import SimpleNamespace
    import pytest
    from geometry_rl_amd.rollout import RolloutDriver
    N, T = 5, 7
    drv = RolloutDriver(updater=None, spec=SimpleNamespace(family="cloth"), seed=2, mini_batch_size=10, sampling="uniform")
    mbs = drv.epoch_minibatches(N, T, torch.device("cpu"))
    assert len(mbs) == (N * T) // 10 and all(m.numel() == 10 for m in mbs)          # short last minibatch dropped
    flat = torch.cat(mbs).tolist()
    assert len(set(flat)) == len(flat) and set(flat) <= set(range(N * T))            # without replacement
    # the guard follows RAGGEDNESS (a per-sample point count among the infos), not the task family: rigid tasks and variable-length ropes
    from geometry_rl_amd import graph
    for spec in (graph.rigid_spec(), graph.rope_spec(variable_length=True)):
        with pytest.raises(ValueError):
            RolloutDriver(updater=None, spec=spec, sampling="uniform")
        RolloutDriver(updater=None, spec=spec, sampling="uniform", allow_stale_topology=True)
    RolloutDriver(updater=None, spec=graph.rope_spec(), sampling="uniform")    # fixed-length ropes and cloth: every sample has the same graph
    RolloutDriver(updater=None, spec=graph.cloth_spec(), sampling="uniform")
