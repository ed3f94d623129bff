"""Data-parallel algebra of the policy update, checked with a real 2-process gloo group on CPU.

Each rank holds half of the minibatch.  With (a) advantage mean/std, (b) the whole-tensor LayerNorm statistics of the critic
all-reduced, and every loss term scaled by 1/B_global, the SUM of the per-rank gradients must equal the single-process
gradient of the full minibatch.  The product's HIP path uses exactly this protocol (geometry_rl_amd/trpl.py, ops.DeepSetsValue,
agent.PolicyUpdater); its GPU counterpart is tests/test_gpu_dp.py."""
import os
import socket

import torch

from spawn_util import spawn_ranks
import torch.distributed as dist
import torch.multiprocessing as mp


def _global_ln_stats(x):
    """whole-tensor mean / biased std over the GLOBAL batch from all-reduced (sum, sum of squares).  The all-reduce is the
    differentiable one: the statistics depend on every rank's activations, so the backward pass all-reduces the matching
    gradient sums (the HIP path does this explicitly with its `bstats` buffers, csrc/critic_ops.hip)."""
    import torch.distributed.nn.functional as dfn
    n = torch.tensor([float(x.numel())], dtype=torch.float64)
    dist.all_reduce(n)
    v = dfn.all_reduce(torch.stack([x.sum(), (x * x).sum()]))
    mean = v[0] / n[0]
    std = (v[1] / n[0] - mean * mean).clamp_min(0).sqrt()
    return mean, std


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import graph as ogr, step as ost
    from geometry_rl_amd import synthetic as syn
    B = 8
    spec = ogr.rigid_spec(P=8, E_mesh=4)
    cfg = ost.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
    a, c = ost.init_agent_params(spec, cfg, seed=3)
    batch = dict(syn.make_rigid_obs(B, P=8, E_mesh=4, seed=1))
    batch.update(syn.make_ppo_fields(B, 6, seed=1))
    # single-process reference on the full minibatch
    full = ost.OracleAgent(spec, cfg, a, c, dtype=torch.float64)
    out = full.loss(batch)
    (out["loss_objective"] + out["loss_entropy"] + out["loss_trust_region"]).backward()
    out["loss_critic"].backward()
    # this rank's shard with all-reduced statistics
    lo, hi = rank * B // world, (rank + 1) * B // world
    shard = {k: v[lo:hi] for k, v in batch.items()}
    adv = batch["advantage"].double()
    st = torch.tensor([shard["advantage"].double().sum(), (shard["advantage"].double() ** 2).sum(), float(hi - lo)], dtype=torch.float64)
    dist.all_reduce(st)
    mean = st[0] / st[2]
    std = ((st[1] - st[2] * mean * mean) / (st[2] - 1)).sqrt().clamp_min(1e-6)
    assert torch.allclose(mean, adv.mean()) and torch.allclose(std, adv.std())
    part = ost.OracleAgent(spec, cfg, a, c, dtype=torch.float64)
    o = part.loss(shard, adv_stats=(mean, std), stats_fn=_global_ln_stats)
    w = (hi - lo) / B  # shard mean -> share of the global mean
    ((o["loss_objective"] + o["loss_entropy"] + o["loss_trust_region"]) * w).backward()
    (o["loss_critic"] * w).backward()
    worst = 0.0
    for params_f, params_p in ((full.actor, part.actor), (full.critic, part.critic)):
        for k, pf in params_f.items():
            if pf.grad is None:
                continue
            g = params_p[k].grad.clone()
            dist.all_reduce(g)  # the one gradient all-reduce of the step
            worst = max(worst, float((g - pf.grad).abs().max() / (1e-12 + pf.grad.abs().max())))
    ret[rank] = worst
    dist.destroy_process_group()


def test_sharded_minibatch_reproduces_full_batch_gradient():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    spawn_ranks(_worker, world, (world,), (ret,))
    assert len(ret) == world
    for r, w in ret.items():
        assert w < 1e-9, (r, w)


def _stats_worker(rank, world, port, ret):
    """rollout.RolloutDriver.publish_advantage_stats: one all-reduce per EPOCH gives every frame the global (sum, sum of squares) of
    the advantages of the minibatch it belongs to (minibatch j of the job = the ranks' minibatches j together)."""
    from types import SimpleNamespace
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from geometry_rl_amd.rollout import RolloutBuffer, RolloutDriver
    N, T = 3, 4
    g = torch.Generator().manual_seed(10 + rank)
    buf = RolloutBuffer({"advantage": torch.randn(N, T, 1, generator=g)})
    calls = []

    def reduce_(kind, t, label=None, lane="m"):
        calls.append(label)
        dist.all_reduce(t)
    upd = SimpleNamespace(group=dist.group.WORLD, loss_module=SimpleNamespace(normalize_advantage=True), _reduce=reduce_)
    drv = RolloutDriver(updater=upd, spec=None, seed=5 + rank)    # every rank shuffles its own environments
    ok = True
    for epoch in range(2):
        idxs = drv.epoch_minibatches(N, T, torch.device("cpu"))
        drv.publish_advantage_stats(buf, idxs)
        for idx in idxs:
            mine = buf.flat("advantage").reshape(-1).double()[idx]
            both = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            a = torch.cat(both)
            want = torch.stack([a.sum(), (a * a).sum()])
            got = buf.flat("adv_stats")[idx]
            ok = ok and got.dtype == torch.float64 and torch.allclose(got, want.expand_as(got), rtol=1e-13, atol=1e-13)
    ret[rank] = (ok, calls)
    dist.destroy_process_group()


def test_epoch_advantage_statistics_two_ranks():
    world = 2
    ret = mp.Manager().dict()
    spawn_ranks(_stats_worker, world, (world,), (ret,))
    for r in range(world):
        ok, calls = ret[r]
        assert ok and calls == ["advantage_stats_epoch"] * 2, (ok, calls)   # ONE collective per epoch
