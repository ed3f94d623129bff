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

# Algorithmic FLOPs per row (row = edge x orientation or node x orientation) of each MFMA kernel, DESIGN.md section 4: the
# products the kernel has to form from the inputs it is given, un-padded (basis width 14, not the 16 the tile carries).
_CHAIN = 2 * 64 * (14 + 64 + 64)             # basis MLP 14->64->64 and the 64->64 kernel layer
FLOPS_PER_ROW = {
    "edge_conv_fwd_kernel": _CHAIN + 2 * 64,                                       # + message multiply, scatter add
    "edge_conv_bwd_x_kernel": _CHAIN + 2 * 64,                                      # recompute, d x_src row (+ sum)
    "edge_conv_bwd_w_kernel": (_CHAIN - 2 * 64 * 64) + 2 * 64 + 2 * 64 * 64 + 2 * (2 * 64 * 64) + 2 * 64 * 64 + 2 * 64 * 14,
    # ^ z1, z2 recompute, dK, dWk, dG2, dG1, dW2, dW1
    "node_mlp_fwd_kernel": 4 * 64 * 256,
    "node_mlp_bwd_fused_kernel": 10 * 64 * 256,                                    # z recompute, dH, dA, dW3, dW4
}
ENTRY_TO_KERNEL = {"grl_edge_conv_fwd": "edge_conv_fwd_kernel", "grl_node_mlp_fwd": "node_mlp_fwd_kernel",
                   "grl_node_mlp_bwd": "node_mlp_bwd_fused_kernel"}
PEAK_F32_MFMA = 157.3          # TFLOP/s, MI355X_MICROARCH.md:42 (the path is specified and checked in f32)
PEAK_BF16X3 = 2500.0 / 3.0     # TFLOP/s of f32-equivalent products when each is three dense bf16 MFMAs (guide: ~2.5 PF dense)


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
    elif name == "rigid2_empn":
        spec = graph.rigid_spec(G=2)
        cfg = agent.AgentConfig(model="empn")  # configs/rigid_insertion_two_agents_multi_empn_trpl_cfg.yaml
        make = lambda B, seed, off: syn.make_rigid_obs(B, G=2, seed=seed, env_offset=off)
        cfg_name = "rigid_insertion_two_agents_multi_empn_trpl"
    else:
        raise ValueError(name)
    return spec, cfg, make, cfg_name


def cpu_baseline(wl_name, minibatch, sample=1024, steps=3, max_threads=32):
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
    ap.add_argument("--pool", type=int, default=8, help="time steps of the synthetic device-resident rollout the minibatches are "
                    "sampled from (without replacement, one frame per env: train.py:128,258); the full 128-step rollout is 1.6 GB")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the recorded hipGraph(s)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    group = None
    if os.environ.get("GRL_BENCH_ONE_GPU"):   # functional test of the N > 1 path on a one-GPU box: every rank on cuda:0, gloo
        local = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if os.environ.get("GRL_BENCH_ONE_GPU"):
            dist.init_process_group("gloo")
        else:
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
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, clip_grad_norm=cfg.clip_grad_norm, max_grad_norm=cfg.max_grad_norm, group=group,
                              use_graph=not args.no_graph)
    # device-resident rollout [B envs, pool steps, ...] + the reference's once-per-rollout work (critic over T+1 frames, shifted GAE)
    from geometry_rl_amd.rollout import RolloutBuffer, RolloutDriver
    T_roll = len(pool)
    data = {k: torch.stack([f[k] for f in pool], dim=1) for k in pool[0]}
    g_in = syn.make_gae_inputs(B, T_roll, seed=rank)
    data.update(reward=g_in["reward"].reshape(B, T_roll, 1).to(dev), done=g_in["done"].reshape(B, T_roll, 1).to(dev),
                terminated=g_in["terminated"].reshape(B, T_roll, 1).to(dev))
    buf = RolloutBuffer(data)
    drv = RolloutDriver(upd, spec, ppo_epochs=5, seed=rank)
    next_last = {k: pool[0][k].unsqueeze(1) for k in spec.in_features}
    torch.cuda.synchronize()
    t_adv = time.perf_counter()
    drv.compute_advantages(buf, next_last)
    torch.cuda.synchronize()
    adv_ms = 1e3 * (time.perf_counter() - t_adv)

    def sampler():   # sampling without replacement, reshuffled every epoch
        while True:
            idx = drv.epoch_indices(B, T_roll, dev)
            for j in range(T_roll):
                yield idx[j]
    mb = sampler()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(3):  # set-up steps (not warmup): the first runs eagerly and builds the cached topology, the second records the
        upd.step_from(buf, next(mb))  # hipGraph(s), the third is the first replay
    for i in range(args.warmup):
        upd.step_from(buf, next(mb))
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = upd.step_from(buf, next(mb))
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
    if rank != 0 and not args.no_roofline:   # the profiled steps are collective when data parallel: every rank runs them
        upd.use_graph = False
        for i in range(6):
            upd.step_from(buf, next(mb))
    if rank == 0 and not args.no_roofline:
        # HIP events around every C-ABI launch, on the launch stream; one profiled step is discarded (first-use cost of
        # timed events lands on a random kernel) and the per-step totals are reduced with the median over the other steps.
        n_prof = 6
        per_step = []
        upd.use_graph = False   # per-kernel HIP events need the launches themselves, not a graph replay (same kernels, same stream)
        for i in range(n_prof):
            hip.KERNEL_TIMES = {}
            hip.KERNEL_ROWS.clear()
            hip.kernel_prof_enable(True)
            upd.step_from(buf, next(mb))
            entry = hip.kernel_time_summary()
            inner = hip.kernel_prof_summary()   # the kernels inside grl_edge_conv_bwd / grl_node_mlp_bwd, one by one
            hip.kernel_prof_enable(False)
            rec = {ENTRY_TO_KERNEL.get(k, k): v for k, v in entry.items() if k != "grl_edge_conv_bwd"}
            rec.update(inner)
            per_step.append(rec)
        hip.KERNEL_TIMES = None
        per_step = per_step[1:]
        n_prof = len(per_step)
        med = lambda xs: sorted(xs)[len(xs) // 2]
        summ = {k: (per_step[0][k][0], med([s_[k][1] for s_ in per_step])) for k in per_step[0]}  # launches/step, ms/step
        rows_of = {"edge_conv_fwd_kernel": "grl_edge_conv_fwd", "edge_conv_bwd_x_kernel": "grl_edge_conv_bwd",
                   "edge_conv_bwd_w_kernel": "grl_edge_conv_bwd", "node_mlp_fwd_kernel": "grl_node_mlp_fwd",
                   "node_mlp_bwd_fused_kernel": "grl_node_mlp_bwd"}
        rows_step = dict(hip.KERNEL_ROWS)      # rows handed to each entry point during the last profiled step
        kernels = {}
        for k, fl in FLOPS_PER_ROW.items():
            if k not in summ:
                continue
            launches, ms_step = summ[k]
            flops_step = fl * rows_step[rows_of[k]]
            ach = flops_step / (ms_step * 1e-3) / 1e12
            kernels[k] = {"launches_per_step": launches, "avg_launch_ms": ms_step / launches, "ms_per_step": ms_step,
                          "rows_per_step": rows_step[rows_of[k]], "gflop_per_launch": flops_step / launches / 1e9, "achieved": ach,
                          "frac": ach / PEAK_F32_MFMA, "frac_of_bf16x3": ach / PEAK_BF16X3}
        name = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        d = kernels[name]
        # HBM traffic per launch: PMC counters cannot be collected from inside this process; the figure comes from the committed
        # summary of the separate `rocprofv3 --pmc` passes (tools/pmc_passes.sh -> profiles/r01_pmc_summary_*.json), same workload
        traffic, traffic_src = None, None
        try:
            import glob
            f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r01_pmc_summary_*.json")))[-1]
            pk = json.load(open(f))["kernels"].get(name)
            if pk:
                traffic = pk["hbm_read_bytes_per_launch"] + pk["hbm_write_bytes_per_launch"]
                traffic_src = os.path.basename(f)
        except Exception:
            pass
        # whole-step figures from SURVEY.md section 8(d): algorithmic FLOPs (3 x forward) and bytes of a perfectly fused step
        step_fig = None
        try:
            topo_b = actor.hyper_data._cache[B]
            O_, C_, W_ = 16, 64, 256
            F_ = O_ * C_ * 4
            n_of = lambda t: topo_b["n_main"] if t == topo_b["main"] else B * topo_b["n_per"][t]
            convs = [(et, topo_b["edges"][et].n_src, topo_b["edges"][et].n_dst, topo_b["edges"][et].n_edges)
                     for rnd in actor.gnn.processor for et, _c in rnd.items() if et in topo_b["edges"]] if hasattr(actor.gnn, "processor") else []
            types_used = {t for et, _, _, _ in convs for t in (et[0], et[2])}
            n_nodes = sum(n_of(t) for t in types_used)
            c_in = len(spec.node_types) + spec.n_vec
            fwd = n_nodes * O_ * c_in * C_ * 2
            fwd += sum(E * O_ * 2 * (14 * C_ + C_ * C_) + O_ * O_ * 2 * (3 * C_ + C_ * C_) for _, _, _, E in convs)
            fwd += sum(E * O_ * C_ * C_ * 2 + 2 * E * O_ * C_ + O_ * O_ * C_ * C_ * 2 + nd * O_ * O_ * C_ * 2 + nd * O_ * 4 * C_ * W_
                       + 8 * nd * O_ * C_ for _, _, nd, E in convs)
            fwd += B * spec.num_actuators * O_ * C_ * (cfg.output_dim + cfg.output_dim_vec) * 2
            n_params = upd.flat.numel()
            obs_bytes = sum(v.numel() * 4 for k, v in pool[0].items() if k in spec.in_features)
            byt = sum(((ns + 2 * nd) + (2 * ns + 3 * nd)) * F_ for _, ns, nd, _ in convs) + 2 * n_nodes * F_ \
                + sum(2 * 16 * E for _, _, _, E in convs) + 2 * obs_bytes + B * (4 * A + A * A + 7) * 4 + 10 * 4 * n_params
            t_step = ms * 1e-3
            step_fig = {"alg_tflop": 3 * fwd / 1e12, "alg_gbyte": byt / 1e9,
                        "mfma_f32_frac": 3 * fwd / t_step / 1e12 / PEAK_F32_MFMA, "hbm_frac": byt / t_step / 8.0e12,
                        "note": "formulas of SURVEY.md 8(d) with the realised node / edge counts of this minibatch; time = the timed region"}
        except Exception as e:  # never let bookkeeping break the benchmark line
            step_fig = {"error": repr(e)}
        roof = {"bound": "mfma", "kernel": name, "achieved": d["achieved"], "peak": PEAK_F32_MFMA, "unit": "TFLOP/s",
                "frac": d["frac"], "traffic": traffic, "traffic_unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)",
                "traffic_source": traffic_src, "avg_launch_ms": d["avg_launch_ms"],
                "launches_per_step": d["launches_per_step"], "gflop_per_launch": d["gflop_per_launch"],
                "peak_note": "f32-exact MFMA peak; the kernel forms each f32 product from three bf16 MFMAs (split-bf16), whose "
                             f"f32-equivalent peak is {PEAK_BF16X3:.0f} TFLOP/s: frac_of_bf16x3",
                "frac_of_bf16x3": d["frac_of_bf16x3"], "whole_step": step_fig, "mfma_kernels": kernels,
                "per_kernel_ms_per_step": {k: v[1] for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "rigid_hepi":   # N = 1 only
        cpu = cpu_baseline(args.workload, args.minibatch)

    if rank == 0:
        line = {
            "metric": "policy-update steps/sec, HEPi 4096 envs x 128 steps", "value": args.steps / dt, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg_name}, 4096 synthetic envs x 128 steps, minibatch {args.minibatch} frames "
                                   f"({B} per GPU), 640 updates per rollout", "global_minibatch": args.minibatch,
                       "parallelism": f"dp{world}"},
            "gae_ms_per_rollout_scan": gae_ms, "advantage_pass_ms": adv_ms,
            "minibatches": f"sampled without replacement from a device-resident {B} x {T_roll}-frame rollout per GPU (one frame per env), "
                           "gathered into the static inputs of the recorded step by one launch",
            "loss": {k: float(out[k].detach()) for k in ("loss_objective", "loss_trust_region", "loss_critic", "kl")},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
