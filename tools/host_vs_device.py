"""Is a small replayed step bound by the host (hipGraphLaunch + Python per step) or by the device?  Per minibatch size: wall time per step
with the device kept busy (the bench's figure), host time per step to ENQUEUE the steps (no synchronisation inside the loop), and the
device-only time of a step (one replay, synchronised, minus an empty-queue replay's launch latency is not separable: reported as is).
   python tools/host_vs_device.py [sizes...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from geometry_rl_amd import agent, synthetic as syn
from geometry_rl_amd.rollout import RolloutBuffer, RolloutDriver

dev = torch.device("cuda:0")
group = None
if "--dp" in sys.argv:   # the data-parallel program on a one-rank RCCL group (GRL_FORCE_DP_PLAN)
    import socket, torch.distributed as dist
    sys.argv.remove("--dp")
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port_ = s_.getsockname()[1]; s_.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    group = dist.group.WORLD
for mb in [int(a) for a in sys.argv[1:]] or [32, 512, 4096]:
    spec, cfg, make_obs, _ = B.workload("rigid_hepi")
    torch.manual_seed(0)
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev, group=group)
    A = spec.num_actuators * cfg.output_dim_vec * 3
    pool = []
    for i in range(8):
        b = dict(make_obs(mb, 100 + i, 0)); b.update(syn.make_ppo_fields(mb, A, seed=i)); pool.append({k: v.to(dev) for k, v in b.items()})
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, use_graph=True, group=group, force_dp_plan=group is not None)
    data = {k: torch.stack([f[k] for f in pool], dim=1) for k in pool[0]}
    buf = RolloutBuffer(data)
    drv = RolloutDriver(upd, spec, ppo_epochs=5, seed=0)
    idx = drv.epoch_indices(mb, 8, dev)
    for i in range(12):
        upd.step_from(buf, idx[i % 8])
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        upd.step_from(buf, idx[i % 8])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    if group is not None:
        print(f"minibatch {mb:5d} (data-parallel program, one-rank RCCL group): step {1e3 * t_all / n:.3f} ms wall, {1e3 * t_host / n:.3f} ms host enqueue")
        continue
    # the graph replay alone (no gather, no Python bookkeeping of step_from)
    g = [p[1] for p in upd._program if p[0] == "graph"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        for gr in g:
            gr.replay()
    t_rep_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_rep_all = time.perf_counter() - t0
    print(f"minibatch {mb:5d}: step {1e3 * t_all / n:.3f} ms wall, {1e3 * t_host / n:.3f} ms host enqueue | graph replay alone "
          f"{1e3 * t_rep_all / n:.3f} ms wall, {1e3 * t_rep_host / n:.3f} ms host")
