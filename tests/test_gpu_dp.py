"""Two data-parallel ranks of the HIP path (both on cuda:0, gloo collectives) must reproduce the single-rank policy update
of the full minibatch: same post-Adam parameters, same loss dict.  Exercises every all-reduce of the step (advantage stats,
critic LayerNorm stats forward/backward, loss sums/maxes, flat gradient)."""
import os
import socket

import pytest
import torch

from spawn_util import spawn_ranks
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _setup(B, group):
    from geometry_rl_amd import agent, graph, synthetic as syn
    dev = torch.device("cuda:0")
    spec = graph.rigid_spec(G=2, angular_velocity=False, object_velocity=False)
    cfg = agent.AgentConfig()
    torch.manual_seed(0)
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev, group=group)
    batch = dict(syn.make_rigid_obs(B, G=2, angular_velocity=False, object_velocity=False, seed=4))
    batch.update(syn.make_ppo_fields(B, 6, seed=4))
    return spec, cfg, actor, critic, loss, {k: v.to(dev) for k, v in batch.items()}


def _worker(rank, world, port, B, ret, use_graph=False, n_steps=1, published=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from geometry_rl_amd import agent
    spec, cfg, actor, critic, loss, batch = _setup(B, dist.group.WORLD)
    with torch.no_grad():  # calibrate on the full batch so every rank starts from identical weights
        actor.forward_diag(*[batch[k] for k in spec.in_features], train=True)
    lo, hi = rank * B // world, (rank + 1) * B // world
    shard = {k: v[lo:hi].contiguous() for k, v in batch.items()}
    if published:   # the minibatch's GLOBAL advantage sums as a per-frame column (rollout.RolloutDriver.publish_advantage_stats)
        a = batch["advantage"].reshape(-1).double()
        shard["adv_stats"] = torch.stack([a.sum(), (a * a).sum()]).expand(hi - lo, 2).contiguous()
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, group=dist.group.WORLD, use_graph=use_graph)
    for i in range(n_steps):
        if i == n_steps - 1:
            upd.collective_log = {}
        out = upd.step(shard)
    ret[f"collectives{rank}"] = sorted(upd.collective_summary(1))
    ret[rank] = ({k: float(out[k].detach()) for k in ("loss_objective", "loss_trust_region", "loss_entropy", "loss_critic", "kl",
                                                     "mean_constraint_max", "ESS")}, upd.flat.detach().cpu())
    dist.destroy_process_group()


def test_two_ranks_match_single_rank():
    from geometry_rl_amd import agent
    B, world = 16, 2
    spec, cfg, actor, critic, loss, batch = _setup(B, None)
    with torch.no_grad():
        actor.forward_diag(*[batch[k] for k in spec.in_features], train=True)
    upd = agent.PolicyUpdater(loss, lr=cfg.lr)
    out = upd.step(batch)
    ref_losses = {k: float(out[k].detach()) for k in ("loss_objective", "loss_trust_region", "loss_entropy", "loss_critic", "kl",
                                                      "mean_constraint_max", "ESS")}
    ref_flat = upd.flat.detach().cpu()
    mgr = mp.Manager()
    ret = mgr.dict()
    spawn_ranks(_worker, world, (world,), (B, ret,))
    assert all(r in ret for r in range(world))
    for r in range(world):
        losses, flat = ret[r]
        for k, v in ref_losses.items():
            assert abs(losses[k] - v) <= 1e-5 * max(1.0, abs(v)), (r, k, losses[k], v)
        err = (flat - ref_flat).abs().max().item()
        print(f"rank {r}: max |param - single-rank param| = {err:.3e}")
        assert err <= 2e-6


def _run_single(B, n_steps, use_graph):
    from geometry_rl_amd import agent
    spec, cfg, actor, critic, loss, batch = _setup(B, None)
    with torch.no_grad():
        actor.forward_diag(*[batch[k] for k in spec.in_features], train=True)
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, use_graph=use_graph)
    for _ in range(n_steps):
        out = upd.step(batch)
    losses = {k: float(out[k].detach()) for k in ("loss_objective", "loss_trust_region", "loss_entropy", "loss_critic", "kl",
                                                  "mean_constraint_max", "ESS")}
    return losses, upd.flat.detach().cpu()


def test_graph_replay_matches_eager():
    """The hipGraph-recorded step (replayed 3 times: the Adam step count lives on the device) equals 3 eager steps."""
    ref_losses, ref_flat = _run_single(16, 3, use_graph=False)
    losses, flat = _run_single(16, 3, use_graph=True)
    for k, v in ref_losses.items():
        assert abs(losses[k] - v) <= 1e-6 * max(1.0, abs(v)), (k, losses[k], v)
    assert (flat - ref_flat).abs().max().item() <= 1e-7


@pytest.mark.parametrize("n_steps", [2, 5])
def test_two_ranks_with_graph_segments_match_single_rank(n_steps):
    """Data parallel with the step recorded as hipGraph segments on two lanes (actor | critic + its collectives) between the
    eager collectives: step 1 eager, step 2 records, later steps replay (5 steps: four replays of both lanes' graphs)."""
    B, world = 16, 2
    ref_losses, ref_flat = _run_single(B, n_steps, use_graph=False)
    mgr = mp.Manager()
    ret = mgr.dict()
    spawn_ranks(_worker, world, (world,), (B, ret, True, n_steps,))
    assert all(r in ret for r in range(world))
    for r in range(world):
        losses, flat = ret[r]
        for k, v in ref_losses.items():
            assert abs(losses[k] - v) <= (1e-5 if n_steps == 2 else 1e-4) * max(1.0, abs(v)), (r, k, losses[k], v)
        assert (flat - ref_flat).abs().max().item() <= (4e-6 if n_steps == 2 else 3e-5)


def test_two_ranks_on_one_communicator_fallback(monkeypatch):
    """GRL_DP_ONE_COMM=1 -- both lanes on the actor's communicator, the documented fallback of PolicyUpdater._plan_dp (what
    tools/first_multigpu_run.sh switches to after a hang): same results as one rank, recorded graph segments included."""
    monkeypatch.setenv("GRL_DP_ONE_COMM", "1")   # (inherited by the spawned workers)
    B, world, n_steps = 16, 2, 3
    ref_losses, ref_flat = _run_single(B, n_steps, use_graph=False)
    mgr = mp.Manager()
    ret = mgr.dict()
    spawn_ranks(_worker, world, (world,), (B, ret, True, n_steps,))
    assert all(r in ret for r in range(world))
    for r in range(world):
        losses, flat = ret[r]
        for k, v in ref_losses.items():
            assert abs(losses[k] - v) <= 1e-4 * max(1.0, abs(v)), (r, k, losses[k], v)
        assert (flat - ref_flat).abs().max().item() <= 3e-5


def test_two_ranks_with_gated_critic_lane(monkeypatch):
    """GRL_DP_GATE_FROM (round 6): the data-parallel program with its critic lane gated behind the actor's first edge convolution (a launch of the
    critic's lane waiting for a flag the actor's fiber convolution raises) -- scheduling only: the same results as one rank, recorded segments included."""
    monkeypatch.setenv("GRL_DP_GATE_FROM", "1")   # (inherited by the spawned workers: every shard size is gated)
    B, world, n_steps = 16, 2, 3
    ref_losses, ref_flat = _run_single(B, n_steps, use_graph=False)
    mgr = mp.Manager()
    ret = mgr.dict()
    spawn_ranks(_worker, world, (world,), (B, ret, True, n_steps,))
    assert all(r in ret for r in range(world))
    for r in range(world):
        losses, flat = ret[r]
        for k, v in ref_losses.items():
            assert abs(losses[k] - v) <= 1e-4 * max(1.0, abs(v)), (r, k, losses[k], v)
        assert (flat - ref_flat).abs().max().item() <= 3e-5


def test_two_ranks_with_published_advantage_statistics():
    """The epoch-level advantage statistics (one all-reduce per epoch, carried as a per-frame column): the update has no
    ``advantage_stats`` collective any more and still equals the single-rank update of the whole minibatch."""
    B, world, n_steps = 16, 2, 3
    ref_losses, ref_flat = _run_single(B, n_steps, use_graph=False)
    mgr = mp.Manager()
    ret = mgr.dict()
    spawn_ranks(_worker, world, (world,), (B, ret, True, n_steps, True,))
    for r in range(world):
        losses, flat = ret[r]
        names = ret[f"collectives{r}"]
        assert "advantage_stats" not in names and "flat_gradient_actor+loss_records" in names and "loss_records" not in names, names
        for k, v in ref_losses.items():
            assert abs(losses[k] - v) <= 2e-5 * max(1.0, abs(v)), (r, k, losses[k], v)
        assert (flat - ref_flat).abs().max().item() <= 8e-6


def _worker_natural(rank, world, port, B, ret, backend="gloo", anneal=False):
    """The natural construction order -- build_agent -> PolicyUpdater -> step(shard) -- with NO manual pre-calibration: the
    data-dependent conv re-initialisation (conv.py:104-105) happens inside the first step, from statistics summed over the ranks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from geometry_rl_amd import agent, graph, synthetic as syn
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    spec = graph.rigid_spec(G=2, angular_velocity=False, object_velocity=False)
    cfg = agent.AgentConfig()
    torch.manual_seed(100 + rank)   # replicas are built from DIFFERENT seeds: rank 0's parameters must win
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev, group=dist.group.WORLD)
    batch = dict(syn.make_rigid_obs(B, G=2, angular_velocity=False, object_velocity=False, seed=4))
    batch.update(syn.make_ppo_fields(B, 6, seed=4))
    lo, hi = rank * B // world, (rank + 1) * B // world
    shard = {k: v[lo:hi].contiguous().to(dev) for k, v in batch.items()}
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, group=dist.group.WORLD, use_graph=True)
    for i in range(4):
        if anneal:
            upd.anneal_lr(cfg.lr, i, 8)
        out = upd.step(shard)
    ret[rank] = ({k: float(out[k].detach()) for k in ("loss_objective", "loss_trust_region", "loss_critic", "kl")},
                 upd.flat.detach().cpu())
    dist.destroy_process_group()


def _single_natural(B, anneal=False):
    from geometry_rl_amd import agent, graph, synthetic as syn
    dev = torch.device("cuda:0")
    spec = graph.rigid_spec(G=2, angular_velocity=False, object_velocity=False)
    cfg = agent.AgentConfig()
    torch.manual_seed(100)
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
    batch = dict(syn.make_rigid_obs(B, G=2, angular_velocity=False, object_velocity=False, seed=4))
    batch.update(syn.make_ppo_fields(B, 6, seed=4))
    batch = {k: v.to(dev) for k, v in batch.items()}
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, use_graph=False)
    for i in range(4):
        if anneal:
            upd.anneal_lr(cfg.lr, i, 8)
        out = upd.step(batch)
    return ({k: float(out[k].detach()) for k in ("loss_objective", "loss_trust_region", "loss_critic", "kl")}, upd.flat.detach().cpu())


@pytest.mark.parametrize("anneal", [False, True])
def test_replicas_calibrate_together_and_follow_the_lr_schedule(anneal):
    """No pre-calibration, replicas initialised from different seeds, the step replayed from hipGraph segments, and (anneal=True)
    the learning rate changed before every step (train.py:264-271): two ranks must land on the single-rank eager parameters."""
    B, world = 16, 2
    ref_losses, ref_flat = _single_natural(B, anneal)
    ret = mp.Manager().dict()
    spawn_ranks(_worker_natural, world, (world,), (B, ret, "gloo", anneal,))
    for r in range(world):
        losses, flat = ret[r]
        for k, v in ref_losses.items():
            assert abs(losses[k] - v) <= 1e-4 * max(1.0, abs(v)), (r, k, losses[k], v)
        err = (flat - ref_flat).abs().max().item()
        print(f"rank {r}: max |param - single-rank param| = {err:.3e}")
        assert err <= 3e-5
    assert torch.equal(ret[0][1], ret[1][1]), "replicas diverged"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs (RCCL over xGMI)")
def test_two_ranks_rccl():
    """The same natural-order run with backend nccl (= RCCL), one rank per GPU -- only where two devices are visible."""
    B, world = 16, 2
    ref_losses, ref_flat = _single_natural(B)
    ret = mp.Manager().dict()
    spawn_ranks(_worker_natural, world, (world,), (B, ret, "nccl",))
    for r in range(world):
        losses, flat = ret[r]
        for k, v in ref_losses.items():
            assert abs(losses[k] - v) <= 1e-4 * max(1.0, abs(v)), (r, k, losses[k], v)
        assert (flat - ref_flat).abs().max().item() <= 3e-5
    assert torch.equal(ret[0][1], ret[1][1]), "replicas diverged"


# ---- the once-per-rollout critic pass under data parallelism: the time-batched launch set with ONE all-reduce per
#      LayerNorm stage for all time steps of a chunk (gnn_vf_net.py:72-80: statistics per time step, here of the WHOLE sharded batch)
def _critic_inputs(N, T):
    from geometry_rl_amd import graph, synthetic as syn
    spec = graph.rigid_spec()
    frames = [syn.make_rigid_obs(N, seed=70 + t) for t in range(T)]
    return spec, {k: torch.stack([f[k] for f in frames], dim=1) for k in spec.in_features}


def _critic_worker(rank, world, port, N, T, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from geometry_rl_amd import agent
    dev = torch.device("cuda:0")
    spec, obs = _critic_inputs(N, T)
    torch.manual_seed(0)
    actor, critic, proj, loss = agent.build_agent(spec, agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2),
                                                  device=dev, group=dist.group.WORLD)
    lo, hi = rank * N // world, (rank + 1) * N // world
    shard = [obs[k][lo:hi].contiguous().to(dev) for k in spec.in_features]
    calls = {"n": 0}
    orig = dist.all_reduce

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    dist.all_reduce = counting
    with torch.no_grad():
        v = critic(*shard, train=False)
    dist.all_reduce = orig
    ret[rank] = (v.reshape(hi - lo, T).cpu(), calls["n"])
    dist.destroy_process_group()


def test_time_batched_critic_pass_two_ranks_equals_one_rank():
    from geometry_rl_amd import agent
    N, T, world = 12, 9, 2
    dev = torch.device("cuda:0")
    spec, obs = _critic_inputs(N, T)
    torch.manual_seed(0)
    actor, critic, proj, loss = agent.build_agent(spec, agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2), device=dev)
    with torch.no_grad():
        ref = critic(*[obs[k].to(dev) for k in spec.in_features], train=False).reshape(N, T).cpu()
    mgr = mp.Manager()
    ret = mgr.dict()
    spawn_ranks(_critic_worker, world, (world,), (N, T, ret,))
    got = torch.cat([ret[r][0] for r in range(world)], dim=0)
    err = (got - ref).abs().max().item()
    print(f"time-batched critic, {world} ranks vs 1: max |dV| = {err:.3e} (|V| max {ref.abs().max().item():.3e}); all-reduces per rank: {ret[0][1]}")
    assert err <= 1e-6 * max(1.0, ref.abs().max().item())
    assert ret[0][1] == 2 and ret[1][1] == 2, "one collective per LayerNorm stage for ALL time steps of the chunk"


def test_loss_records_ride_on_a_float_sum():
    """grl_trpl_fold_record_pairs / grl_trpl_report_record_pairs: a rank's record as (hi, lo) float pairs in its own row of a
    [world][14] region, zeros elsewhere -- the float SUM of the ranks' regions (what the gradient's all-reduce does to it) holds every
    record to ~2^-48, and the values reported from it equal the ones from the all-gathered fp64 records."""
    from geometry_rl_amd import hip
    dev = torch.device("cuda:0")
    world, batch = 3, 100
    nb = (batch + 15) // 16
    g = torch.Generator().manual_seed(0)
    regions, recs = [], []
    for r in range(world):
        slots = (torch.randn(nb, 14, generator=g, dtype=torch.float64) * 10 ** torch.randint(-3, 4, (nb, 14), generator=g).double()).abs()
        slots[:, 10] = 16.0                              # (column 10 counts the frames)
        slots = slots.to(dev).contiguous()
        rec = torch.empty(14, device=dev, dtype=torch.float64)
        hip.call("grl_trpl_fold_record", slots, batch, rec)
        region = torch.full((world * 28,), 7.0, device=dev, dtype=torch.float32)   # stale content must be overwritten
        hip.call("grl_trpl_fold_record_pairs", slots, batch, r, world, region)
        pairs = region.view(world, 14, 2).double()
        assert torch.equal(pairs[torch.arange(world) != r], torch.zeros(world - 1, 14, 2, device=dev, dtype=torch.float64))
        back = pairs[r, :, 0] + pairs[r, :, 1]
        assert ((back - rec).abs() <= 2.0 ** -46 * rec.abs()).all(), (back, rec)
        regions.append(region)
        recs.append(rec)
    summed = regions[2] + (regions[0] + regions[1])      # any order: adding zeros is exact
    assert torch.equal(summed, regions[0] + regions[1] + regions[2])
    out_a, out_b = torch.empty(14, device=dev), torch.empty(14, device=dev)
    sums_a, sums_b = torch.empty(12, device=dev, dtype=torch.float64), torch.empty(12, device=dev, dtype=torch.float64)
    mx_a, mx_b = torch.empty(2, device=dev, dtype=torch.int32), torch.empty(2, device=dev, dtype=torch.int32)
    hip.call("grl_trpl_report_records", torch.stack(recs).contiguous(), world, sums_a, mx_a, 0.01, out_a)
    hip.call("grl_trpl_report_record_pairs", summed, world, sums_b, mx_b, 0.01, out_b)
    assert torch.equal(mx_a, mx_b)
    assert ((sums_a - sums_b).abs() <= 1e-12 * sums_a.abs()).all()
    assert torch.allclose(out_a, out_b, rtol=1e-6, atol=0)
