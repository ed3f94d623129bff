"""Debug helper: per-launch HIP-event times of the C-ABI calls of a few policy-update steps (compare with rocprofv3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import agent, graph, hip, synthetic as syn
dev = torch.device("cuda:0")
spec = graph.rigid_spec()
cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
B = 4096
b = dict(syn.make_rigid_obs(B, seed=1)); b.update(syn.make_ppo_fields(B, 6, seed=1))
b = {k: v.to(dev) for k, v in b.items()}
with torch.no_grad():
    actor.forward_diag(*[b[k] for k in spec.in_features], train=True)
upd = agent.PolicyUpdater(loss)
for _ in range(3):
    upd.step(b)
torch.cuda.synchronize()
pool = [b]
for i in range(3):
    c = dict(syn.make_rigid_obs(B, seed=100 + i)); c.update(syn.make_ppo_fields(B, 6, seed=i))
    pool.append({k: v.to(dev) for k, v in c.items()})
hip.KERNEL_TIMES = {}
for i in range(4):
    upd.step(pool[i])
torch.cuda.synchronize()
for k, v in hip.KERNEL_TIMES.items():
    if "edge" in k or "node_mlp" in k:
        print(k, [round(a.elapsed_time(c), 3) for a, c in v])
