"""Where should the critic's lane start?  One rank, two lanes: the critic's small kernels co-run with whatever the actor's lane is executing;
beside the first edge convolution they cost that launch ~60-90 us at 4096 frames (DESIGN.md finding 42).  This sweeps an idle delay in front
of the critic's lane (PolicyUpdater.critic_delay_us -> grl_calib_spin) and prints the step time per delay, alternating, on one box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from geometry_rl_amd import agent, synthetic as syn
dev = torch.device("cuda:0")
wl = os.environ.get("GRL_WORKLOAD", "rigid_hepi")
B = int(os.environ.get("GRL_B", "4096"))
spec, cfg, make_obs, _ = bench.workload(wl)
A = spec.num_actuators * cfg.output_dim_vec * 3
delays = [int(x) for x in os.environ.get("GRL_DELAYS", "0 150 400 550 900 1300").split()]
upds = {}
batch = dict(make_obs(B, 1, 0)); batch.update(syn.make_ppo_fields(B, A, seed=1))
batch = {k: v.to(dev) for k, v in batch.items()}
for d in delays:
    torch.manual_seed(0)
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
    u = agent.PolicyUpdater(loss, lr=cfg.lr, clip_grad_norm=cfg.clip_grad_norm, max_grad_norm=cfg.max_grad_norm, use_graph=True)
    u.critic_delay_us = d
    for _ in range(6):
        u.step(batch)
    upds[d] = u
torch.cuda.synchronize()
for rnd in range(3):
    for d in delays:
        u = upds[d]
        for _ in range(5):
            u.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            u.step(batch)
        torch.cuda.synchronize()
        print(f"round {rnd} delay {d:5d} us : {1e3 * (time.perf_counter() - t0) / 60:.4f} ms/step", flush=True)
