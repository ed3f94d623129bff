"""How fast can the HOST issue recorded policy-update steps?  For each minibatch size: K steps enqueued back to back without a device
synchronisation (host seconds per step = the enqueue loop alone), then the synchronised total (device-paced seconds per step).  A shard-sized
step whose host time reaches its device time is launch-bound: nothing on the device makes it faster.
   GRL_WORKLOAD (rigid_hepi), GRL_SIZES ("32 256 512 1024 4096"), GRL_K (300);  also prints cProfile's top entries of the enqueue loop at the first size"""
import cProfile, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from geometry_rl_amd import agent, synthetic as syn
dev = torch.device("cuda:0")
spec, cfg, make_obs, _ = bench.workload(os.environ.get("GRL_WORKLOAD", "rigid_hepi"))
K = int(os.environ.get("GRL_K", "300"))
first = True
for B in [int(x) for x in os.environ.get("GRL_SIZES", "32 256 512 1024 4096").split()]:
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
    b = dict(make_obs(B, 1, 0)); b.update(syn.make_ppo_fields(B, spec.num_actuators * cfg.output_dim_vec * 3, seed=1))
    b = {k: v.to(dev) for k, v in b.items()}
    with torch.no_grad():
        actor.forward_diag(*[b[k] for k in spec.in_features], train=True)
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, clip_grad_norm=cfg.clip_grad_norm, max_grad_norm=cfg.max_grad_norm, use_graph=True)
    for _ in range(6):
        upd.step(b)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            upd.step(b)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
    print(f"{B:5d} frames: host enqueue {min(r[0] for r in res):7.1f} us/step, synchronised {min(r[1] for r in res):7.1f} us/step  ({upd.mode})", flush=True)
    if first:
        first = False
        pr = cProfile.Profile(); pr.enable()
        for _ in range(K):
            upd.step(b)
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
