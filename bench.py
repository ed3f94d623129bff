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
    # the fused backward (edge_conv16.hip, default since round 2): ONE chain recompute, d x_src row, dK, dWk, dG2, dW2, dG1, dW1
    "edge_bwd16_kernel": _CHAIN + 2 * 64 + 2 * 64 + 4 * (2 * 64 * 64) + 2 * 64 * 14,
    "node_mlp_fwd_kernel": 4 * 64 * 256,
    "node_mlp_bwd16_kernel": 10 * 64 * 256,                                        # z recompute, dH, dA, dW3, dW4
}
# Algorithmic HBM bytes per row of the same kernels with fp32 latents (bf16 latents: half): the row gathers of the edge
# kernels (x_src | x_src + dM; the per-node stores, 256 B / in-degree, are left out) and the row streams of the node kernels
BYTES_PER_ROW = {"edge_conv_fwd_kernel": 256, "edge_bwd16_kernel": 512, "node_mlp_fwd_kernel": 768, "node_mlp_bwd16_kernel": 768}
# the second roofline convention (a backward pass is priced at 2 x its forward, no credit for the chain / z recompute the
# fused kernels execute): `frac_alg_3xfwd` beside `frac` (= executed FLOPs) in the line and per kernel; forward kernels: the same figure
FLOPS_PER_ROW_3XFWD = {"edge_conv_fwd_kernel": _CHAIN + 2 * 64, "edge_bwd16_kernel": 2 * (_CHAIN + 2 * 64),
                       "node_mlp_fwd_kernel": 4 * 64 * 256, "node_mlp_bwd16_kernel": 2 * (4 * 64 * 256)}
# (entry names are normalised before the lookup: _bf16 / _balanced / _img suffixes dropped)
ENTRY_TO_KERNEL = {"grl_edge_conv_fwd": "edge_conv_fwd_kernel",
                   "grl_node_mlp_fwd": "node_mlp_fwd_kernel",
                   "grl_node_mlp_bwd": "node_mlp_bwd16_kernel"}
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
    elif name in ("rope_hepi_var", "rope_hepi_bf16"):
        # BASELINE config 5: variable-length rope graphs (80- and 40-link ropes mixed in every minibatch); "_bf16": one bf16 MFMA per
        # dense product in the actor (fp32 accumulation and storage), tolerance tests/test_gpu_bf16_rope.py
        spec = graph.rope_spec(variable_length=True)
        cfg = agent.AgentConfig(dim=2, clip_grad_norm=True, precision="bf16" if name.endswith("bf16") else "fp32")
        make = lambda B, seed, off: syn.make_rope_obs(B, seed=seed, variable_length=True, env_offset=off)
        cfg_name = "rope_shaping_hepi_trpl (variable-length ropes" + (", bf16 products)" if name.endswith("bf16") else ")")
    elif name == "rigid2_empn":
        spec = graph.rigid_spec(G=2)
        cfg = agent.AgentConfig(model="empn")  # configs/rigid_insertion_two_agents_multi_empn_trpl_cfg.yaml
        make = lambda B, seed, off: syn.make_rigid_obs(B, G=2, seed=seed, env_offset=off)
        cfg_name = "rigid_insertion_two_agents_multi_empn_trpl"
    else:
        raise ValueError(name)
    return spec, cfg, make, cfg_name


LOSS_KEYS = ("loss_objective", "loss_trust_region", "loss_entropy", "loss_critic", "ESS", "kl", "constraint", "mean_constraint",
             "mean_constraint_max", "cov_constraint", "cov_constraint_max", "entropy", "entropy_diff")


# Oracle twin of every bench workload and the bounded sample its CPU leg runs on: frames per oracle update (sized so that the parity
# update + the timed oracle updates stay within ~10-40 s of CPU work on the GPU box's host cores) and the tolerance BASELINE.md section 3
# states for the workload (1e-4 for the fp32 builds; 2e-2 relative for config 5's bf16 build)
ORACLE_CASES = {
    "rigid_hepi":     dict(spec="rigid", spec_kw={}, sample=1024, tol=1e-4, ptol=2e-5),
    "cloth_hepi":     dict(spec="cloth", spec_kw={}, sample=256, tol=1e-4, ptol=2e-5),
    "rope_hepi":      dict(spec="rope", spec_kw={}, sample=128, tol=1e-4, ptol=2e-5),
    "rope_hepi_var":  dict(spec="rope", spec_kw=dict(variable_length=True), sample=128, tol=1e-4, ptol=2e-5),
    "rope_hepi_bf16": dict(spec="rope", spec_kw=dict(variable_length=True), sample=128, tol=2e-2, ptol=None),
    "rigid2_empn":    dict(spec="rigid", spec_kw=dict(G=2), sample=512, tol=1e-4, ptol=2e-5),
}


# the committed PMC summaries `roofline.traffic` may be copied from (tools/pmc_passes.sh + tools/pmc_report.py), by workload
PMC_SUMMARY = {"rigid_hepi": "r06_pmc_summary_rigid_hepi.json", "rope_hepi_bf16": "r06_pmc_summary_rope_hepi_bf16.json"}


def cpu_baseline_and_parity(wl_name, minibatch, dev, steps=3, max_threads=32):
    """(1) BASELINE.md section 3's parity gate: one update of a bounded-size minibatch of THIS workload through the HIP path and through
    the oracle (CPU restatement of the reference path) from identical parameters and inputs -- loc, var, state_value and every
    loss-dict entry within tol * max(1, |ref|), post-Adam parameters within 2e-5 (fp32 builds; the bf16 build of config 5 is held
    to its own 2e-2 bar and its parameters are not compared: the oracle is fp32);
    (2) the oracle timed on this box's host cores on that bounded sample (the reported cpu_baseline, kind "port")."""
    import dataclasses
    from oracle import graph as ogr, step as ost
    from geometry_rl_amd import agent
    case = ORACLE_CASES[wl_name]
    sample, tol, ptol = case["sample"], case["tol"], case["ptol"]
    spec, cfg, make_obs, _ = workload(wl_name)
    o_spec = getattr(ogr, case["spec"] + "_spec")(**case["spec_kw"])
    o_fields = {f.name for f in dataclasses.fields(ost.AgentConfig)}
    kw = {k: v for k, v in dataclasses.asdict(cfg).items() if k in o_fields and k != "precision"}
    o_cfg = ost.AgentConfig(**kw)
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, max_threads)  # more intra-op threads than this only slow the small CPU ops of the oracle down
    torch.set_num_threads(cores)
    A = spec.num_actuators * cfg.output_dim_vec * 3
    a, c = ost.init_agent_params(o_spec, o_cfg, seed=0)
    ag = ost.OracleAgent(o_spec, o_cfg, a, c)
    batch = dict(make_obs(sample, 1, 0))
    batch.update(syn_fields(sample, A))
    with torch.no_grad():
        ag.actor_forward({k: batch[k] for k in o_spec.in_features}, calibrate=True)
    # HIP twin from the oracle's calibrated parameters
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
    actor.load_state_dict({k: v.detach().to(dev) for k, v in ag.actor.items()}, strict=False)
    critic.load_state_dict({"_network1." + k: v.detach().to(dev) for k, v in ag.critic.items()}, strict=False)
    for m in actor.modules():
        if hasattr(m, "callibrated"):
            m.callibrated.fill_(True)
    actor._calib_checked = True
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, clip_grad_norm=cfg.clip_grad_norm, max_grad_norm=cfg.max_grad_norm)
    out = upd.step({k: v.to(dev) for k, v in batch.items()})
    ref, ref_grads = ag.update(batch)      # also the warm-up of the timed loop below
    worst, worst_key, ok = 0.0, None, True
    pairs = [("loc", out["loc"], ref["loc"]), ("var", out["sigma"] ** 2, ref["var"]), ("state_value", out["state_value"], ref["state_value"])]
    pairs += [(k, out[k], ref[k]) for k in LOSS_KEYS]
    floor = 1.0 if tol <= 1e-3 else 1e-2   # the 2e-2 bar of the bf16 build is relative down to 1e-2 (tests/test_gpu_bf16_rope.py)
    for k, g, r in pairs:
        r = torch.as_tensor(r).detach().double().cpu()
        e = (torch.as_tensor(g).detach().double().cpu().reshape(r.shape) - r).abs().max().item() / max(floor if r.numel() == 1 else 1.0, r.abs().max().item())
        if not (e <= tol):
            ok = False
        if e > worst or e != e:
            worst, worst_key = e, k
    p_err, p_ratio = None, None
    if ptol is not None:
        # post-Adam parameters, the tests' rule (oracle/parity_util.py): the first Adam step turns an absolute gradient error dg into
        # lr * dg * eps / (|g| + eps)^2 -- entries with a large gradient are saturated and held to fp32 rounding, only entries near zero get
        # the lr * dg / eps allowance -- with dg = 2e-4 of the tensor's own largest reference gradient entry (at least 1e-4 of the network's).
        # ENTRY-WISE wherever nothing rescales the gradient; per tensor under clip_grad_norm (the clip coefficient carries the relative
        # error of the norm on top).  (the per-tensor form saturates at 2 lr for every tensor with max |g| >= 0.1.)
        from oracle import parity_util as pu
        p_err, p_ratio = 0.0, 0.0
        for mod, ref_p, ref_g, strip in ((actor, ag.actor, ref_grads["actor"], 0), (critic, ag.critic, ref_grads["critic"], len("_network1."))):
            scales = pu.grad_scales(ref_g)
            for k, p in mod.named_parameters():
                kk = k[strip:]
                if kk not in ref_g:   # no gradient reaches it (the unused `_mean` head): untouched by both optimizers
                    allowed = pu.P_ROUND * max(1.0, float(ref_p[kk].abs().max()))
                elif cfg.clip_grad_norm:
                    allowed = pu.adam_first_step_bound(cfg.lr, 1e-5, scales[kk], clip=True, p_ref=ref_p[kk])
                else:
                    allowed = pu.adam_first_step_bound_elem(cfg.lr, 1e-5, ref_g[kk], scales[kk], ref_p[kk])
                e = (p.detach().cpu() - ref_p[kk]).abs().max().item()
                p_err, p_ratio = max(p_err, e), max(p_ratio, pu.param_excess(p, ref_p[kk], allowed))
        ok = ok and p_ratio <= 1.0
    gate = {"passed": bool(ok), "workload": wl_name, "frames": sample,
            "tolerance": f"{tol:g} * max(1, |ref|) on loc / var / state_value / 13 loss entries" + ("; post-Adam parameters ENTRY-WISE within lr * min(2, dg eps / (max(|g_i| - dg, 0) + eps)^2) + 4e-7 max(1, |p|), dg = 2e-4 * max|g_tensor| (oracle/parity_util.py; per tensor under clip_grad_norm)" if ptol else
                         " (relative down to 1e-2 for the scalar entries; parameters not compared: bf16 build against the fp32 oracle)"),
            "worst_value_err_over_scale": worst, "worst_key": worst_key, "post_adam_param_err": p_err,
            "post_adam_param_err_over_allowed": p_ratio}
    t0 = time.perf_counter()
    for _ in range(steps):
        ag.update(batch)
    dt = (time.perf_counter() - t0) / steps
    cpu = {"value": (sample / minibatch) / dt, "unit": "policy-update steps/s", "cores": cores, "host_cores": host_cores, "kind": "port",
           "sample": f"{steps} oracle updates of a {sample}-frame minibatch of this workload ({dt:.2f} s each), scaled linearly to {minibatch} frames",
           "torch_threads": torch.get_num_threads()}
    return cpu, gate


def determinism_selfcheck(dev, prec=""):
    """Two identical launches of each MFMA op (edge convolution and ConvNeXt block, forward + backward, 8192 nodes / 24576 edges) must agree
    BITWISE (the one-wave kernels rely on unfenced MFMA groups; DESIGN.md finding 15 records one pool box that once failed this).
    Run once per process before anything is timed; a failure voids the line."""
    from geometry_rl_amd import hip, ops, hepi
    g = torch.Generator().manual_seed(0)
    n, E = 8192, 24576
    ei = torch.stack([torch.randint(0, n, (E,), generator=g), torch.arange(n).repeat_interleave(3)])
    es = ops.build_edge_set(ei.to(dev), n, n)
    dt = hip.storage_dtype(prec)
    rnd = lambda *s_, sc=1.0: (torch.randn(*s_, generator=g) * sc).to(dev)
    x, dy, xd = rnd(n, 16, 64).to(dt), rnd(n, 16, 64).to(dt), rnd(n, 16, 64).to(dt)
    ps = torch.rand(n, 3, generator=g).to(dev)
    grid = hepi.make_grid(3, 16).to(dev).contiguous()
    ew = [rnd(64, 14, sc=0.25), rnd(64), rnd(64, 64, sc=0.125), rnd(64), rnd(64, 64, sc=0.125)]
    mw = [1 + rnd(64, sc=0.1), rnd(64, sc=0.1), rnd(256, 64, sc=0.125), rnd(256, sc=0.1), rnd(64, 256, sc=0.06), rnd(64, sc=0.1)]

    def run():
        xs = x.clone().requires_grad_(True)
        ws = [w.clone().requires_grad_(True) for w in ew]
        y = ops.EdgeConv.apply(xs, ps, ps, grid, *ws, es, 3, None, prec)
        y.backward(dy)
        x2 = x.clone().requires_grad_(True)
        ms = [w.clone().requires_grad_(True) for w in mw]
        z = ops.NodeMLP.apply(x2, xd, *ms, None, None, prec)
        z.backward(dy)
        return [y.detach(), xs.grad, z.detach(), x2.grad] + [w.grad for w in ws + ms]

    a, b = run(), run()
    torch.cuda.synchronize()
    bad = [i for i, (u, v) in enumerate(zip(a, b)) if not torch.equal(u, v)]
    return {"passed": not bad, "differing_outputs": bad, "shape": f"{n} nodes, {E} edges, edge conv + ConvNeXt block forward / backward twice"}


# Reference box for `value_normalised`: the calibration figures of the box the round-5 target table was measured on.
# value_normalised = value * REF_BOX["mfma_tflops"] / this box's mfma_tflops -- the 4096-frame step is 84 % MFMA kernels whose rate follows
# the clock the chip holds under matrix load (DESIGN.md finding 32), which is what the MFMA calibration loop measures.
REF_BOX = {"mfma_tflops": 1750.0, "copy_tb_per_s": 4.8}   # (boxes seen in round 5: 1717-1782 TFLOP/s, 4.78-4.86 TB/s)


def box_calibration(dev, mfma_iters=6000, reps=9, copy_bytes=1 << 30):
    """Two FIXED kernels (csrc/calib.hip), timed with HIP events in this process before anything else runs: a bare v_mfma_f32_32x32x16_bf16
    loop on random register operands on every SIMD, and a 1 GiB float4 copy.  They identify the box: the pool's boxes span several per
    cent for one build, more than the changes of a round.  Median over ``reps`` back-to-back launches after two untimed ones."""
    from geometry_rl_amd import hip
    out = torch.empty(256 * 256, device=dev, dtype=torch.float32)
    med = lambda xs: sorted(xs)[len(xs) // 2]

    def timed(fn, n):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in ev]

    f_mfma = lambda: hip.call("grl_calib_mfma", mfma_iters, out)
    timed(f_mfma, 2)
    t_m = timed(f_mfma, reps)
    flops = 1024.0 * mfma_iters * 16 * 32768
    src = torch.empty(copy_bytes // 4, device=dev, dtype=torch.float32).normal_()
    dst = torch.empty_like(src)
    import ctypes
    f_copy = lambda: hip.call("grl_calib_copy", src, dst, ctypes.c_longlong(copy_bytes))
    timed(f_copy, 2)
    t_c = timed(f_copy, reps)
    del src, dst
    torch.cuda.empty_cache()
    mf, cp = flops / (med(t_m) * 1e-3) / 1e12, 2.0 * copy_bytes / (med(t_c) * 1e-3) / 1e12
    return {"mfma_tflops": mf, "mfma_tflops_min_max": [flops / (max(t_m) * 1e-3) / 1e12, flops / (min(t_m) * 1e-3) / 1e12],
            "mfma_ms_per_launch": med(t_m), "copy_tb_per_s": cp, "copy_ms_per_launch": med(t_c),
            "reference_box": REF_BOX, "mfma_vs_reference": mf / REF_BOX["mfma_tflops"], "copy_vs_reference": cp / REF_BOX["copy_tb_per_s"],
            "what": f"grl_calib_mfma: 1024 waves x {mfma_iters} x 16 v_mfma_f32_32x32x16_bf16 on random register operands (dense bf16 peak 2500); "
                    f"grl_calib_copy: {copy_bytes >> 20} MiB float4 copy, read + write bytes; medians of {reps} launches, HIP events"}


def syn_fields(B, A):
    from geometry_rl_amd import synthetic as syn
    return syn.make_ppo_fields(B, A, seed=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=7, help="the timed region of --steps steps is repeated this many times back to back; "
                    "`value` is the MEDIAN repeat (every repeat is listed in the line)")
    ap.add_argument("--workload", default="rigid_hepi")
    ap.add_argument("--minibatch", type=int, default=4096, help="global frames per policy update (= num_envs)")
    ap.add_argument("--pool", type=int, default=128, help="time steps of the synthetic device-resident rollout the minibatches are "
                    "sampled from (without replacement, one frame per env: train.py:128,258); 128 = the whole 4096 x 128 rollout "
                    "(1.6 GB of observations in HBM)")
    ap.add_argument("--no-parity-gate", action="store_true", help="skip the 1024-frame oracle comparison (and the CPU baseline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--critic-gate", default=None, help="A/B: where the critic's lane starts: edge0 (default) | fiber0 | fwd_end")
    ap.add_argument("--no-critic-gate", action="store_true", help="A/B: the critic's lane starts with the step instead of behind the actor's first edge convolution")
    ap.add_argument("--one-stream", action="store_true", help="A/B: one rank, everything on ONE stream / in ONE hipGraph (PolicyUpdater(overlap_critic=False)): no cross-stream wait anywhere")
    ap.add_argument("--unroll", type=int, default=8, help="one rank: minibatch steps per recorded launch (PolicyUpdater.run_minibatches); 1 = one step per launch, as in round 5")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the recorded hipGraph(s)")
    ap.add_argument("--dp-plan", action="store_true", help="one GPU, but the DATA-PARALLEL program of the step: a one-rank RCCL process group, "
                    "every collective issued, graph segments between them -- what a shard's step costs before any inter-GPU latency")
    args = ap.parse_args()

    # ---- `python bench.py --gpus N` outside a launcher: start the N ranks ourselves.  Nothing above has touched the GPU (no HIP call,
    #      no torch.cuda query), so the child launcher and its ranks are ordinary fresh processes; this process only relays their
    #      output and exit code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    group = None
    if os.environ.get("GRL_BENCH_ONE_GPU"):   # functional test of the N > 1 path on a one-GPU box: every rank on cuda:0, gloo
        local = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        # a collective that never completes must end the run, not hold the node: the watchdog aborts after three minutes (the default is ten
        # to thirty) -- nothing of the N > 1 path has ever run on more than one GPU
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("GRL_BENCH_COLLECTIVE_TIMEOUT_S", "180")))
        if os.environ.get("GRL_BENCH_ONE_GPU"):
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=tmo)
        group = dist.group.WORLD
    elif args.dp_plan:
        import socket
        import torch.distributed as dist
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port_ = s_.getsockname()[1]; s_.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        group = dist.group.WORLD
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from geometry_rl_amd import agent, hip, synthetic as syn
    spec, cfg, make_obs, cfg_name = workload(args.workload)
    calib = None
    if rank == 0:   # before anything else has warmed or loaded the chip (a library without the calibration kernels -- an older build loaded
        try:        # through GRL_LIB for an A/B -- simply has no calibration block)
            calib = box_calibration(dev)
        except AttributeError:
            calib = None
    det = None
    if rank == 0 and not os.environ.get("GRL_BENCH_NO_SELFCHECK"):   # (the override is for timing knock-out builds, whose results are garbage)
        det = determinism_selfcheck(dev, "_bf16" if cfg.precision == "bf16" else "")
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
    # the natural order: the updater is built first; the data-dependent calibration (conv.py:104-105) happens inside its first step,
    # from statistics summed over the ranks (every replica computes the factors of the whole minibatch)
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, clip_grad_norm=cfg.clip_grad_norm, max_grad_norm=cfg.max_grad_norm, group=group,
                              use_graph=not args.no_graph, force_dp_plan=args.dp_plan, overlap_critic=not args.one_stream, critic_after_first_conv=(False if args.no_critic_gate else (args.critic_gate or True)))
    if world > 1:
        import torch.distributed as dist
        assert dist.get_world_size() == world
    # device-resident rollout [B envs, pool steps, ...] + the reference's once-per-rollout work (critic over T+1 frames, shifted GAE)
    from geometry_rl_amd.rollout import RolloutBuffer, RolloutDriver
    T_roll = len(pool)
    data = {k: torch.stack([f[k] for f in pool], dim=1) for k in pool[0]}
    first_frame = {k: pool[0][k].clone() for k in pool[0]}
    pool = None   # the frames now live in `data` only
    g_in = syn.make_gae_inputs(B, T_roll, seed=rank)
    data.update(reward=g_in["reward"].reshape(B, T_roll, 1).to(dev), done=g_in["done"].reshape(B, T_roll, 1).to(dev),
                terminated=g_in["terminated"].reshape(B, T_roll, 1).to(dev))
    buf = RolloutBuffer(data)
    drv = RolloutDriver(upd, spec, ppo_epochs=5, seed=rank)
    next_last = {k: first_frame[k].unsqueeze(1) for k in spec.in_features}
    adv_times = []
    for _ in range(2):   # first call: cold (topology of the rollout-sized batch, lazy initialisation); second: what every later rollout pays
        torch.cuda.synchronize()
        t_adv = time.perf_counter()
        drv.compute_advantages(buf, next_last)
        torch.cuda.synchronize()
        adv_times.append(1e3 * (time.perf_counter() - t_adv))
    adv_ms_cold, adv_ms = adv_times

    def sampler():   # sampling without replacement, reshuffled every epoch
        while True:
            idx = drv.epoch_indices(B, T_roll, dev)
            drv.publish_advantage_stats(buf, list(idx))   # N > 1: the epoch's advantage statistics by ONE all-reduce (rollout.py)
            for j in range(T_roll):
                yield idx[j]
    mb = sampler()

    def take(n):   # the next n minibatches as an [n, B_local] index matrix (what an epoch's sampler hands out, train.py:258-261)
        return torch.stack([next(mb) for _ in range(n)])
    # one rank, recorded lanes: the minibatches of a chunk go to the device as several steps per launch (PolicyUpdater.run_minibatches);
    # every other form (data parallel, eager, one-stream, --unroll 1) takes one step per call
    chunked = world == 1 and not args.dp_plan and args.unroll > 1

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(3):  # set-up steps (not warmup): the first runs eagerly and builds the cached topology, the second records the
        upd.step_from(buf, next(mb))  # hipGraph(s), the third is the first replay
    if chunked:   # (the warm-up also records the multi-step launch)
        if getattr(upd, "autotune_form", False):   # set-up, not warm-up: the updater measures once which recorded form this size takes
            upd.run_minibatches(buf, take(upd.tune_minibatches(args.unroll)), unroll=args.unroll)
        upd.run_minibatches(buf, take(max(args.warmup, args.unroll)), unroll=args.unroll)
    else:
        for i in range(args.warmup):
            upd.step_from(buf, next(mb))
    # R repeats of the timed region, each EXACTLY --steps steps between a barrier + device synchronisation on both sides; the repeat's
    # time is the MAX over the ranks; `value` is the median repeat (one 65 ms shot could not resolve a 1 % change)
    rep_dt, rep_enq = [], []
    for r_ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        if chunked:
            out = upd.run_minibatches(buf, take(args.steps), unroll=args.unroll)
        else:
            for i in range(args.steps):
                out = upd.step_from(buf, next(mb))
        rep_enq.append(time.perf_counter() - t0)   # the HOST's share: all K steps enqueued (no device synchronisation inside the loop)
        barrier()
        rep_dt.append(time.perf_counter() - t0)
    if world > 1:
        import torch.distributed as dist
        t_all = torch.tensor(rep_dt, device=dev, dtype=torch.float64)
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        rep_dt_global = [float(x) for x in t_all.tolist()]
    else:
        rep_dt_global = list(rep_dt)
    dt_own = sorted(rep_dt)[len(rep_dt) // 2]
    dt = sorted(rep_dt_global)[len(rep_dt_global) // 2]
    # The same MFMA calibration once more, HOT, directly behind the timed region: a box that throttles under sustained load reads its normal
    # figure at process start and loses 20 % a few seconds into the workload (round 5 met one: profiles/r05_sizes_and_workloads_v6_throttled_box.txt).
    calib_after = None
    if rank == 0 and calib is not None:
        from geometry_rl_amd import hip as _hip
        _out = torch.empty(256 * 256, device=dev, dtype=torch.float32)
        _ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a_, b_ in _ev:
            a_.record()
            _hip.call("grl_calib_mfma", 6000, _out)
            b_.record()
        torch.cuda.synchronize()
        _t = sorted(a_.elapsed_time(b_) for a_, b_ in _ev)[2]
        _mf = 1024.0 * 6000 * 16 * 32768 / (_t * 1e-3) / 1e12
        calib_after = {"mfma_tflops": _mf, "vs_start_of_process": _mf / calib["mfma_tflops"],
                       "box_throttled": bool(_mf < 0.93 * calib["mfma_tflops"]),
                       "what": "grl_calib_mfma again, median of 5 launches directly behind the timed region; box_throttled: below 0.93 of the figure at process start"}
    dp_info = None
    if world > 1 or args.dp_plan:
        import torch.distributed as dist
        own = torch.tensor([dt_own], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(every, own)
        per_rank = [1e3 * float(x.item()) / args.steps for x in every]   # every rank's own median repeat
        # the first hardware run must describe itself: which backend carried the collectives, how many ranks it saw,
        # every collective of the step by name with its payload and the time its lane was held by it (HIP events on the lane's stream around
        # the call: waiting for the other ranks + transfer), how long the main lane stood at the two joins with the critic's lane, and the
        # ranks' own step times (max / min) -- five extra steps with the log on, all ranks in lock-step, behind the timed region
        n_log = 5
        upd.collective_log = {}
        for i in range(n_log):
            upd.step_from(buf, next(mb))
        coll = upd.collective_summary(n_log)
        upd.collective_log = None
        try:
            lib_version = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None
        except Exception:
            lib_version = None
        gname = lambda g_: getattr(g_, "group_name", None) if g_ is not None else None
        dp_info = {"collective_backend": dist.get_backend(), "collective_library_version": lib_version, "ranks_seen": dist.get_world_size(),
                   # which communicator carries which lane, and every NCCL_* / RCCL_* setting of the run (algorithm / protocol sweeps of
                   # tools/first_multigpu_run.sh are told apart by these)
                   "communicators": {"actor_lane": gname(upd.group), "critic_lane": gname(upd.group_c) or gname(upd.group),
                                     "two_communicators": upd.group_c is not None},
                   "collective_env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "GRL_DP_"))},
                   "per_rank_ms_per_step": per_rank, "rank_spread_max_over_min": max(per_rank) / max(min(per_rank), 1e-9),
                   "collectives_per_step": sum(v["per_step"] for k, v in coll.items() if not k.startswith(("join", "wait"))),
                   "collectives": coll,
                   "main_lane_waits_ms": {k: v["mean_ms"] for k, v in coll.items() if k.startswith(("join", "wait"))},
                   "note": "rank 0's lanes; mean / max over 5 logged steps behind the timed region; the actor's lane waits for ONE collective, "
                           "'flat_gradient_actor+loss_records' (the ranks' loss records ride in front of the gradient slice), the critic's lane (own communicator) for its four LayerNorm-statistic "
                           "reductions, its gradient slice and its loss sum; the advantage statistics of all minibatches of an epoch are "
                           "reduced once per epoch ('advantage_stats_epoch', outside the update)"}
    ms = 1e3 * dt / args.steps
    n_graphs = sum(1 for p_ in (upd._program or []) if p_[0] == "graph")
    lane_form = getattr(upd, "form_by_size", {}).get(B)
    if chunked and getattr(upd, "_epoch", None) is not None and lane_form != "per_step":
        upd.mode_timed = (f"graph ({upd._epoch['key'][1]} minibatch step(s) per launch: 2 single-stream hipGraphs on two lanes, in-graph gathers "
                          f"{'by device cursor, lanes joined once per call' if upd._epoch['key'][2] else 'of fixed index rows'}, gate as a launch)")
    else:
        upd.mode_timed = upd.mode + ("" if upd.mode != "graph" else
                                     f" ({'one hipGraph' if n_graphs == 1 else str(n_graphs) + ' single-stream hipGraphs on two lanes' if (world == 1 and not args.dp_plan) else str(n_graphs) + ' hipGraph segments between the collectives'})")

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
        # ... and ONE lane: with the critic's lane beside them (as in the timed region) the events of an MFMA launch also time whatever critic
        # kernels happen to share the device with it -- eagerly issued, those land somewhere else than in the replayed step, and the
        # figure moved 0.229-0.249 from box to box.  The roofline fraction is the kernel's own: the profiled steps run the one-stream plan
        # (the same kernels in the same order; `replayed_launches` below are the launches of the timed program, lanes and all).
        lanes_one_rank = world == 1 and not args.dp_plan
        if lanes_one_rank:
            upd.overlap_critic = False
        for i in range(n_prof):
            hip.KERNEL_TIMES = {}
            hip.KERNEL_ROWS.clear()
            hip.kernel_prof_enable(True)
            upd.step_from(buf, next(mb))
            entry = hip.kernel_time_summary()
            inner = hip.kernel_prof_summary()   # the kernels inside grl_edge_conv_bwd / grl_node_mlp_bwd, one by one
            hip.kernel_prof_enable(False)
            entry = {k.replace("_bf16", "").replace("_balanced", "").replace("_img", ""): v for k, v in entry.items()}
            rec = {ENTRY_TO_KERNEL.get(k, k): v for k, v in entry.items() if k != "grl_edge_conv_bwd"}
            rec.update(inner)
            per_step.append(rec)
        hip.KERNEL_TIMES = None
        if lanes_one_rank:
            upd.overlap_critic = True
        per_step = per_step[1:]
        n_prof = len(per_step)
        med = lambda xs: sorted(xs)[len(xs) // 2]
        summ = {k: (per_step[0][k][0], med([s_[k][1] for s_ in per_step])) for k in per_step[0]}  # launches/step, ms/step
        rows_of = {"edge_conv_fwd_kernel": "grl_edge_conv_fwd", "edge_conv_bwd_x_kernel": "grl_edge_conv_bwd",
                   "edge_conv_bwd_w_kernel": "grl_edge_conv_bwd", "edge_bwd16_kernel": "grl_edge_conv_bwd",
                   "node_mlp_fwd_kernel": "grl_node_mlp_fwd",
                   "node_mlp_bwd16_kernel": "grl_node_mlp_bwd"}
        rows_step = {k.replace("_bf16", "").replace("_balanced", "").replace("_img", ""): v for k, v in hip.KERNEL_ROWS.items()}   # rows handed to each entry point (last profiled step)
        pipe_peak = 2500.0 if cfg.precision == "bf16" else PEAK_BF16X3
        kernels = {}
        for k, fl in FLOPS_PER_ROW.items():
            if k not in summ:
                continue
            launches, ms_step = summ[k]
            flops_step = fl * rows_step[rows_of[k]]
            ach = flops_step / (ms_step * 1e-3) / 1e12
            gbs = BYTES_PER_ROW.get(k, 0) * (0.5 if cfg.precision == "bf16" else 1.0) * rows_step[rows_of[k]] / (ms_step * 1e-3) / 1e9
            kernels[k] = {"launches_per_step": launches, "avg_launch_ms": ms_step / launches, "ms_per_step": ms_step,
                          "rows_per_step": rows_step[rows_of[k]], "gflop_per_launch": flops_step / launches / 1e9, "achieved": ach,
                          "frac": ach / PEAK_F32_MFMA, "frac_of_bf16x3": ach / pipe_peak, "alg_gbyte_per_s": gbs, "frac_of_hbm": gbs / 8000.0,
                          "frac_alg_3xfwd": ach / pipe_peak * FLOPS_PER_ROW_3XFWD.get(k, fl) / fl}
        # every MFMA kernel the profiled step ran must have been priced (a mapping lost in the literal above dropped
        # node_mlp_fwd_kernel silently out of `mfma_kernels` and out of the choice of the dominant kernel)
        lost = [k for k in summ if k in ("grl_edge_conv_fwd", "grl_node_mlp_fwd", "grl_node_mlp_bwd", "grl_edge_conv_bwd")]
        assert not lost and all(k in kernels for k in FLOPS_PER_ROW if k in summ), ("unpriced MFMA kernels", lost, sorted(summ))
        # The same MFMA launches as the TIMED region runs them -- replayed from the recorded hipGraph: the step is recorded once more with
        # wall-clock stamp kernels in front of and behind those launches (in-library, grl_prof_enable(2); ordinary kernel nodes, re-run by
        # every replay) and the stamps of 8 replays are read back.  Reported beside the HIP-event figures, which stay the line's `frac`.
        replayed = None
        if world == 1 and not args.no_graph and upd.mode.startswith("graph"):
            try:
                hip.kernel_prof_enable(2)
                upd.use_graph, upd._program, upd._static = True, None, None
                upd.step_from(buf, next(mb))
                reps = []
                for i in range(8):
                    upd.step_from(buf, next(mb))
                    torch.cuda.synchronize()
                    reps.append(hip.kernel_prof_summary())
                replayed = {}
                for k in reps[0]:
                    if k in kernels:
                        n_, t_ = reps[0][k][0], med([r_[k][1] for r_ in reps])
                        fl_ = FLOPS_PER_ROW[k] * rows_step[rows_of[k]]
                        replayed[k] = {"launches_per_step": n_, "avg_launch_ms": t_ / n_, "ms_per_step": t_,
                                       "frac_of_bf16x3": fl_ / (t_ * 1e-3) / 1e12 / pipe_peak}
            except Exception as e:   # bookkeeping must not break the line
                replayed = {"error": repr(e)}
            finally:
                hip.kernel_prof_enable(False)
                upd._program, upd._static = None, None
        name = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        d = kernels[name]
        # HBM traffic per launch: PMC counters cannot be collected from inside this process; the figure comes from the committed
        # summary of the separate `rocprofv3 --pmc` passes (tools/pmc_passes.sh -> profiles/r01_pmc_summary_*.json), same workload
        traffic, traffic_src = None, None
        try:
            # ONE named file per workload (no glob, no "newest"), and only while it describes the library that is loaded: the summary records
            # the source hash of the library the passes ran on (tools/pmc_report.py); a mismatch leaves `traffic` null
            f = os.path.join(ROOT, "profiles", PMC_SUMMARY.get(args.workload, f"pmc_summary_{args.workload}.json"))
            rec_ = json.load(open(f))
            from geometry_rl_amd import hip as _hip
            have = _hip.embedded_hash(_hip.LIB_PATH)
            if rec_.get("source_hash") and rec_["source_hash"] == have:
                pk = rec_["kernels"].get(name)
                if pk:
                    traffic = pk["hbm_read_bytes_per_launch"] + pk["hbm_write_bytes_per_launch"]
                    traffic_src = os.path.basename(f)
            else:
                traffic_src = f"{os.path.basename(f)} NOT used: it describes library {rec_.get('source_hash')}, the loaded one is {have}"
        except Exception:
            pass
        # whole-step figures from SURVEY.md section 8(d): algorithmic FLOPs (3 x forward) and bytes of a perfectly fused step
        step_fig = None
        try:
            topo_b = actor.hyper_data._cache[B]
            O_, C_, W_ = 16, 64, 256
            F_ = O_ * C_ * (2 if cfg.precision == "bf16" else 4)   # bytes of one node's latent block
            n_of = lambda t: topo_b["n_main"] if t == topo_b["main"] else B * topo_b["n_per"][t]
            if hasattr(actor.gnn, "processor"):   # HEPi: one conv per (round, edge type)
                convs = [(et, topo_b["edges"][et].n_src, topo_b["edges"][et].n_dst, topo_b["edges"][et].n_edges)
                         for rnd in actor.gnn.processor for et, _c in rnd.items() if et in topo_b["edges"]]
            else:   # EMPN: every layer runs over the merged edge set (ponita_gcn.py:102-126); the LAST layer is only evaluated where the
                    # read-out reads it (edges into the actuators, node block on the actuators) -- the work counted here is the work needed
                mg = next(v[1] for v in actor.gnn._merged_cache.values() if v[0] is topo_b["edges"])
                main_et = next(iter(topo_b["edges"]))
                L = len(actor.gnn.ponita.interaction_layers)
                convs = [(main_et, mg["n"], mg["n"], mg["es_all"].n_edges)] * (L - 1 if actor.gnn.prune_last_layer else L)
                if actor.gnn.prune_last_layer:
                    convs.append((main_et, mg["n"], mg["n_ro"], mg["es_last"].n_edges))
            types_used = {t for et, _, _, _ in convs for t in (et[0], et[2])} if hasattr(actor.gnn, "processor") else set(topo_b["n_per"])
            n_nodes = sum(n_of(t) for t in types_used if t in topo_b["n_per"])
            c_in = len(spec.node_types) + spec.n_vec
            fwd = n_nodes * O_ * c_in * C_ * 2
            fwd += sum(E * O_ * 2 * (14 * C_ + C_ * C_) + O_ * O_ * 2 * (3 * C_ + C_ * C_) for _, _, _, E in convs)
            fwd += sum(E * O_ * C_ * C_ * 2 + 2 * E * O_ * C_ + O_ * O_ * C_ * C_ * 2 + nd * O_ * O_ * C_ * 2 + nd * O_ * 4 * C_ * W_
                       + 8 * nd * O_ * C_ for _, _, nd, E in convs)
            fwd += B * spec.num_actuators * O_ * C_ * (cfg.output_dim + cfg.output_dim_vec) * 2
            n_params = upd.flat.numel()
            obs_bytes = sum(v.numel() * 4 for k, v in first_frame.items() if k in spec.in_features)
            byt = sum(((ns + 2 * nd) + (2 * ns + 3 * nd)) * F_ for _, ns, nd, _ in convs) + 2 * n_nodes * F_ \
                + sum(2 * 16 * E for _, _, _, E in convs) + 2 * obs_bytes + B * (4 * A + A * A + 7) * 4 + 10 * 4 * n_params
            t_step = ms * 1e-3
            step_fig = {"alg_tflop": 3 * fwd / 1e12, "alg_gbyte": byt / 1e9,
                        "pipe_frac": 3 * fwd / t_step / 1e12 / pipe_peak, "mfma_f32_frac": 3 * fwd / t_step / 1e12 / PEAK_F32_MFMA,
                        "hbm_frac": byt / t_step / 8.0e12,
                        "note": "formulas of SURVEY.md 8(d) with the realised node / edge counts of this minibatch; time = the timed region"}
        except Exception as e:  # never let bookkeeping break the benchmark line
            step_fig = {"error": repr(e)}
        if cfg.precision == "bf16":   # config 5's bf16 build is priced against the HBM roof, not the matrix pipe
            head = {"bound": "hbm", "kernel": name, "achieved": d["alg_gbyte_per_s"], "peak": 8000.0, "unit": "GB/s", "frac": d["frac_of_hbm"],
                    "mfma_frac_of_bf16_pipe": d["frac_of_bf16x3"]}
        else:
            head = {"bound": "mfma", "kernel": name, "achieved": d["achieved"], "peak": pipe_peak, "unit": "TFLOP/s", "frac": d["frac_of_bf16x3"],
                    "frac_alg_3xfwd": d["frac_alg_3xfwd"],
                    "frac_note": "frac = EXECUTED algorithmic FLOPs (the fused backward's one chain recompute included: 52 992 per edge-row) / "
                                 "launch time / peak; frac_alg_3xfwd = the same launch priced by SURVEY.md 8(d)'s rule (backward = 2 x forward = "
                                 "36 608 per edge-row, no recompute credit)"}
        roof = {**head, "traffic": traffic, "traffic_unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)",
                "traffic_source": traffic_src, "avg_launch_ms": d["avg_launch_ms"],
                "launches_per_step": d["launches_per_step"], "gflop_per_launch": d["gflop_per_launch"],
                "avg_launch_note": "HIP events on the launch stream around the eagerly issued launches of the profiled steps behind the timed region, "
                                   "run on ONE stream (no critic kernels beside the timed launch: the kernel's own duration); `replayed_launches` = "
                                   "the same launches inside the replayed two-lane step (wall-clock stamp kernels in the graph); rocprofv3 "
                                   "--kernel-trace of this command (profiles/) averages over both kinds",
                "peak_note": "peak = the pipe the kernel runs on: every f32 product is three dense bf16 MFMAs (split-bf16, f32 "
                             f"accumulate), 2500 / 3 = {PEAK_BF16X3:.0f} TFLOP/s of f32-equivalent products (MI355X_MICROARCH.md: ~2.5 PF "
                             "dense bf16); achieved = algorithmic f32 FLOP per launch / HIP-event launch time",
                "frac_of_f32_mfma_peak": d["frac"], "f32_mfma_peak": PEAK_F32_MFMA, "whole_step": step_fig, "mfma_kernels": kernels,
                "replayed_launches": replayed,
                "replayed_note": "the same launches replayed from the recorded hipGraph (as in the timed region), bracketed by device wall-clock "
                                 "stamp kernels inside the graph (includes ~2-5 us of dispatch per launch); median of 8 replays",
                "per_kernel_ms_per_step": {k: v[1] for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])}}

    cpu, gate = None, None
    if rank == 0 and world == 1 and not (args.no_cpu_baseline or args.no_parity_gate):   # N = 1 only
        cpu, gate = cpu_baseline_and_parity(args.workload, args.minibatch, dev)

    if rank == 0:
        line = {
            "metric": "policy-update steps/sec, HEPi 4096 envs x 128 steps", "value": args.steps / dt, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "repeats": len(rep_dt_global), "repeats_ms_per_step": [1e3 * x / args.steps for x in rep_dt_global],
            # host time to ENQUEUE a step (median repeat, rank 0): a step whose enqueue time reaches its device time is launch-bound
            "host_enqueue_ms_per_step": 1e3 * sorted(rep_enq)[len(rep_enq) // 2] / args.steps,
            # which recorded form the one-rank updater took at this size and the measurement behind it (PolicyUpdater._tune_form; null: the table)
            "lane_form": ({"form": lane_form, **upd.form_times.get(B, {})} if lane_form else None),
            "ms_per_step_min_max": [1e3 * min(rep_dt_global) / args.steps, 1e3 * max(rep_dt_global) / args.steps],
            "value_note": "value = steps / (median over the repeats of the time of one timed region of exactly `steps` steps, max over ranks)",
            "box_calibration": calib, "box_calibration_after": calib_after,
            "value_normalised": (args.steps / dt) / calib["mfma_vs_reference"] if calib else None,
            "value_normalised_note": "value x reference box's MFMA calibration / this box's (BASELINE.md section 4): compare lines of different boxes by this",
            "scaling": "strong", "vs_baseline": None,
            "dtype": ("bf16 latent storage and MFMA operands (one bf16 MFMA per product), f32 accumulate / weights / loss" if cfg.precision == "bf16" else
                      "f32 storage/accumulate, bf16x3 products (three bf16 MFMAs per f32 product)"), "data": "synthetic",
            "mode": upd.mode_timed, "host_cores": os.cpu_count(),
            "config": {"workload": f"{cfg_name}, 4096 synthetic envs x 128 steps, minibatch {args.minibatch} frames "
                                   f"({B} per GPU), 640 updates per rollout", "global_minibatch": args.minibatch,
                       "parallelism": f"dp{world}" + (" (data-parallel program on a one-rank RCCL group: --dp-plan)" if args.dp_plan else "")},
            "gae_ms_per_rollout_scan": gae_ms, "advantage_pass_ms": adv_ms, "advantage_pass_ms_cold": adv_ms_cold,
            "advantage_pass": f"critic over the {T_roll} + 1 frames of all {B} environments per GPU + shifted GAE, once per rollout (train.py:249-251); "
                              "warm = second call; the time steps are groups of one launch set (per-step LayerNorm statistics"
                              + ("" if world == 1 else ", summed over the ranks by one all-reduce per LayerNorm stage and chunk") + ")",
            "minibatches": f"sampled without replacement from a device-resident {B} x {T_roll}-frame rollout per GPU (one frame per env), "
                           "gathered into the static inputs of the recorded step by one launch",
            "loss": {k: float(out[k].detach()) for k in ("loss_objective", "loss_trust_region", "loss_critic", "kl")},
            "roofline": roof, "cpu_baseline": cpu, "parity_gate": gate, "determinism_selfcheck": det,
            "data_parallel": dp_info,
        }
        if det is not None and not det["passed"]:   # a box (or a build) whose MFMA kernels are not reproducible reports no number
            line["invalid_value"], line["value"] = line["value"], None
            print(json.dumps(line))
            raise SystemExit("determinism self-check failed: " + json.dumps(det))
        if gate is not None and not gate["passed"]:   # no speed number without parity
            line["invalid_value"], line["value"] = line["value"], None
            print(json.dumps(line))
            raise SystemExit("parity gate failed: " + json.dumps(gate))
        print(json.dumps(line))
    if world > 1 or args.dp_plan:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
