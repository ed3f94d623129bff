#!/usr/bin/env python3
"""Policy-update throughput of the HIP path on synthetic HEPi rollouts (BASELINE.json metric).

  python bench.py --gpus 1 --steps 50 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One *step* = one minibatch policy update exactly as examples/torchrl/train.py:258-316 performs it: TRPL loss forward
(HEPi actor + DeepSets critic + projection), actor and critic backward, gradient all-reduce when N > 1, two Adam updates.
Workload: rigid_insertion_multi_hepi_trpl, 4096 envs x 128 steps, minibatch = 4096 frames (one frame per env), fp32.
Multi-GPU: the 4096-frame minibatch is sharded over the ranks (strong scaling), one RCCL all-reduce of the flat gradient.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs per row (row = edge x orientation or node x orientation), DESIGN.md "Roofline"
FLOPS_PER_ROW = {
    "grl_edge_conv_fwd": 2 * 64 * (16 + 64 + 64),
    "grl_edge_conv_bwd": 2 * 64 * (16 + 64 + 64) + 2 * 64 * 64 * 4 + 2 * 64 * 16,
    "grl_node_mlp_fwd": 4 * 64 * 256,
    "grl_node_mlp_bwd": 10 * 64 * 256,
}


def workload(name):
    from geometry_rl_amd import agent, graph, synthetic as syn
    if name == "rigid_hepi":
        spec = graph.rigid_spec()
        cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)  # configs/rigid_insertion_multi_hepi_trpl_cfg.yaml:110-116
        make = lambda B, seed, off: syn.make_rigid_obs(B, seed=seed, env_offset=off)
        cfg_name = "rigid_insertion_multi_hepi_trpl"
    elif name == "cloth_hepi":
        spec = graph.cloth_spec()
        cfg = agent.AgentConfig(trust_region_coeff=4.0, cov_bound=0.001)  # configs/cloth_hanging_multi_hepi_trpl_cfg.yaml:130-133
        make = lambda B, seed, off: syn.make_cloth_obs(B, seed=seed)
        cfg_name = "cloth_hanging_multi_hepi_trpl"
    elif name == "rope_hepi":
        spec = graph.rope_spec()
        cfg = agent.AgentConfig(dim=2, clip_grad_norm=True)
        make = lambda B, seed, off: syn.make_rope_obs(B, seed=seed)
        cfg_name = "rope_shaping_hepi_trpl"
    else:
        raise ValueError(name)
    return spec, cfg, make, cfg_name


def cpu_baseline(wl_name, minibatch, sample=128, steps=2, max_threads=32):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores on a bounded sample."""
    from oracle import graph as ogr, step as ost
    from geometry_rl_amd import synthetic as syn
    assert wl_name == "rigid_hepi"
    cores = min(os.cpu_count() or 1, max_threads)  # more intra-op threads than this only slow the small CPU ops down
    torch.set_num_threads(cores)
    spec = ogr.rigid_spec()
    cfg = ost.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
    a, c = ost.init_agent_params(spec, cfg, seed=0)
    ag = ost.OracleAgent(spec, cfg, a, c)
    batch = dict(syn.make_rigid_obs(sample, seed=1))
    batch.update(syn.make_ppo_fields(sample, 6, seed=1))
    with torch.no_grad():
        ag.actor_forward({k: batch[k] for k in spec.in_features}, calibrate=True)
    ag.update(batch)
    t0 = time.perf_counter()
    for _ in range(steps):
        ag.update(batch)
    dt = (time.perf_counter() - t0) / steps
    return {"value": (sample / minibatch) / dt, "unit": "policy-update steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} oracle updates of a {sample}-frame minibatch ({dt:.2f} s each), scaled linearly to {minibatch} frames",
            "torch_threads": torch.get_num_threads()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="rigid_hepi")
    ap.add_argument("--minibatch", type=int, default=4096, help="global frames per policy update (= num_envs)")
    ap.add_argument("--pool", type=int, default=4, help="distinct synthetic minibatches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    group = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        group = dist.group.WORLD
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from geometry_rl_amd import agent, hip, synthetic as syn
    spec, cfg, make_obs, cfg_name = workload(args.workload)
    assert args.minibatch % world == 0
    B = args.minibatch // world
    torch.manual_seed(0)
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev, group=group)
    A = spec.num_actuators * cfg.output_dim_vec * 3
    pool = []
    for i in range(args.pool):
        obs = make_obs(B, 100 + i, rank * B)
        b = dict(obs)
        b.update(syn.make_ppo_fields(B, A, seed=1000 * rank + i))
        pool.append({k: v.to(dev) for k, v in b.items()})
    with torch.no_grad():  # first training call: data-dependent calibration (conv.py:104-105) on rank-local data, then broadcast
        actor.forward_diag(*[pool[0][k] for k in spec.in_features], train=True)
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, clip_grad_norm=cfg.clip_grad_norm, max_grad_norm=cfg.max_grad_norm, group=group)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        upd.step(pool[i % len(pool)])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = upd.step(pool[i % len(pool)])
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = 1e3 * dt / args.steps

    # ---- GAE + shifted critic pass over the whole 4096 x 128 rollout (once per 640 updates; outside the timed region)
    gae_ms = None
    if rank == 0:
        N, T = args.minibatch, 128
        g = syn.make_gae_inputs(N, T, seed=0)
        gd = {k: v.to(dev) for k, v in g.items()}
        agent.gae(gd["reward"], gd["done"], gd["terminated"], gd["values"])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            agent.gae(gd["reward"], gd["done"], gd["terminated"], gd["values"])
        torch.cuda.synchronize()
        gae_ms = 1e3 * (time.perf_counter() - t1) / 5

    roof = None
    if rank == 0 and not args.no_roofline:
        # HIP events around every C-ABI launch, on the launch stream; one profiled step is discarded (first-use cost of
        # timed events lands on a random kernel) and the per-step totals are reduced with the median over the other steps.
        n_prof = 6
        per_step = []
        for i in range(n_prof):
            hip.KERNEL_TIMES = {}
            upd.step(pool[i % len(pool)])
            per_step.append(hip.kernel_time_summary())
        hip.KERNEL_TIMES = None
        per_step = per_step[1:]
        n_prof = len(per_step)
        names = per_step[0].keys()
        med = lambda xs: sorted(xs)[len(xs) // 2]
        summ = {k: (per_step[0][k][0] * n_prof, med([s[k][1] for s in per_step]) * n_prof) for k in names}
        dom = max(summ.items(), key=lambda kv: kv[1][1])
        name, (calls, total) = dom
        # rows processed by the dominant kernel per launch (from the cached topology of this minibatch size)
        topo = actor.hyper_data._cache[B]
        rows = {}
        for et, es in topo["edges"].items():
            rows[et] = es.n_edges * 16
        e_rows = sum(rows.values())
        n_rows = (topo["n_main"] + B * spec.num_actuators * (2 if spec.num_actuators > 1 else 1)) * 16
        per_step_rows = e_rows if "edge" in name else n_rows
        launches_per_step = calls / n_prof
        flops_per_launch = FLOPS_PER_ROW.get(name, 0) * per_step_rows / max(launches_per_step, 1)
        avg_ms = total / calls
        ach = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        roof = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3,
                "traffic": None, "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step,
                "per_kernel_ms_per_step": {k: v[1] / n_prof for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])}}

    cpu = None
    if rank == 0 and not args.no_cpu_baseline and args.workload == "rigid_hepi":
        cpu = cpu_baseline(args.workload, args.minibatch)

    if rank == 0:
        line = {
            "metric": "policy-update steps/sec, HEPi 4096 envs x 128 steps", "value": args.steps / dt, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg_name}, 4096 synthetic envs x 128 steps, minibatch {args.minibatch} frames "
                                   f"({B} per GPU), 640 updates per rollout", "global_minibatch": args.minibatch,
                       "parallelism": f"dp{world}"},
            "gae_ms_per_rollout_scan": gae_ms,
            "loss": {k: float(out[k].detach()) for k in ("loss_objective", "loss_trust_region", "loss_critic", "kl")},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
