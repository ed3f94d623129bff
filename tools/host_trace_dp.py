import os, sys, time, socket
import torch, torch.distributed as dist
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"])
import bench as B
from geometry_rl_amd import agent, synthetic as syn
from geometry_rl_amd.rollout import RolloutBuffer, RolloutDriver
dev = torch.device("cuda:0")
s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port_ = s_.getsockname()[1]; s_.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
group = dist.group.WORLD
mb = 4096
spec, cfg, make_obs, _ = B.workload("rigid_hepi")
torch.manual_seed(0)
actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev, group=group)
A = spec.num_actuators * cfg.output_dim_vec * 3
pool = []
for i in range(4):
    b = dict(make_obs(mb, 100 + i, 0)); b.update(syn.make_ppo_fields(mb, A, seed=i)); pool.append({k: v.to(dev) for k, v in b.items()})
upd = agent.PolicyUpdater(loss, lr=cfg.lr, use_graph=True, group=group, force_dp_plan=group is not None)
data = {k: torch.stack([f[k] for f in pool], dim=1) for k in pool[0]}
buf = RolloutBuffer(data)
drv = RolloutDriver(upd, spec, ppo_epochs=5, seed=0)
idx = drv.epoch_indices(mb, 4, dev)
for i in range(8):
    upd.step_from(buf, idx[i % 4])
torch.cuda.synchronize()
orig = upd._do
log = []
def traced(kind, item, label=None, lane="m"):
    t0 = time.perf_counter()
    orig(kind, item, label, lane)
    log.append((kind, label, lane, 1e6 * (time.perf_counter() - t0)))
upd._do = traced
for i in range(3):
    log.clear()
    t0 = time.perf_counter()
    upd.step_from(buf, idx[i % 4])
    tot = 1e6 * (time.perf_counter() - t0)
print("host us per entry (last step), total", round(tot))
for e in log: print("  %-8s %-28s %s %8.1f" % (e[0], e[1], e[2], e[3]))
