"""`bench.py --gpus 2` end to end, as the driver will launch it on an 8-GPU node -- here with both ranks on the one visible GPU and gloo
(GRL_BENCH_ONE_GPU=1: the collectives are host-staged, the data-parallel program -- graph segments between the all-reduces, lock-step
profiling leg on rank != 0, max-over-ranks timing, rank 0's JSON line -- is the one that runs over RCCL).  A fresh child process: nothing
of this pytest process's GPU state is inherited."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]     # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_rank_bench_line():
    two = _bench("--gpus", "2", "--minibatch", "1024", "--steps", "3", "--warmup", "1", "--pool", "4",
                 env_extra={"GRL_BENCH_ONE_GPU": "1"})
    assert two["n_gpus"] == 2 and two["steps"] == 3 and two["warmup"] == 1
    assert two["scaling"] == "strong" and two["config"]["parallelism"] == "dp2"
    assert "segments" in two["mode"], two["mode"]                    # hipGraph segments between the collectives
    assert two["cpu_baseline"] is None and two["parity_gate"] is None   # N = 1 only
    assert two["roofline"] is not None and two["roofline"]["frac"] > 0     # the profiling leg ran in lock-step on both ranks
    for k, v in two["loss"].items():
        assert v == v and abs(v) < 1e6, (k, v)
    # the N > 1 line describes its collectives: backend, ranks, every collective by name with count / payload / time
    dp = two["data_parallel"]
    assert dp["collective_backend"] == "gloo" and dp["ranks_seen"] == 2 and len(dp["per_rank_ms_per_step"]) == 2
    assert dp["rank_spread_max_over_min"] >= 1.0
    names = set(dp["collectives"])
    assert "advantage_stats" not in names, names      # once per EPOCH (advantage_stats_epoch, rollout.publish_advantage_stats), not per update
    for want in ("critic_ln1_fwd_stats", "critic_ln2_fwd_stats", "critic_ln2_bwd_stats", "critic_ln1_bwd_stats",
                 "flat_gradient_actor+loss_records", "flat_gradient_critic", "loss_critic_sum", "join_critic_lane"):
        assert want in names, (want, names)
        assert dp["collectives"][want]["per_step"] == 1.0 and dp["collectives"][want]["mean_ms"] >= 0.0
    # actor's lane: ONE collective -- its slice of the flat gradient with the ranks' loss records riding in front of it; critic's lane (own
    # communicator): 4 LayerNorm statistics, its slice of the gradient, its loss sum; (+ the epoch's advantage statistics when an epoch
    # starts inside the logged steps)
    assert dp["collectives"]["flat_gradient_actor+loss_records"]["bytes"] > 500_000 and 7.0 <= dp["collectives_per_step"] <= 7.5, dp["collectives_per_step"]
    assert two["advantage_pass_ms"] > 0 and "one all-reduce per LayerNorm stage" in two["advantage_pass"]
    one = _bench("--gpus", "1", "--minibatch", "512", "--steps", "20", "--warmup", "3", "--pool", "4", "--no-parity-gate", "--no-roofline")
    assert one["n_gpus"] == 1 and "hipGraph" in one["mode"] and one["mode"].startswith("graph")
    # Two 512-frame shards time-share ONE GPU here: every one of the step's 7 collectives is host-staged by gloo AND needs the GPU to
    # switch between the two processes' contexts before the other rank's contribution exists (measured on different boxes: 4.6, 10.8,
    # 22, 67, 190, 500 and 937 ms per step against 0.72 ms for a lone 512-frame rank; per-collective log of a 170 ms step: 0.3-0.6 ms
    # for seven of them, 60 ms mean / 303 ms max for one -- a property of this stand-in and of the box's scheduler, not of the program:
    # over RCCL each rank owns its GPU, and the same program on a one-rank RCCL group costs 5 us more than the one-rank step,
    # profiles/r04_dp_plan.txt).  So the bound only separates "the data-parallel program ran" from a hang:
    assert two["ms_per_step"] < 5000.0, (two["ms_per_step"], one["ms_per_step"])
    print("2 ranks x 512 frames on one GPU (gloo):", two["ms_per_step"], "ms/step; 1 rank x 512 frames:", one["ms_per_step"], "ms/step")
