"""A few policy-update steps of the bench workload, for rocprofv3 (kernel trace or PMC passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import agent, graph, hip, synthetic as syn
dev = torch.device("cuda:0")
spec = graph.rigid_spec()
cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
B = int(os.environ.get("GRL_B", "4096"))
b = dict(syn.make_rigid_obs(B, seed=1)); b.update(syn.make_ppo_fields(B, 6, seed=1))
b = {k: v.to(dev) for k, v in b.items()}
with torch.no_grad():
    actor.forward_diag(*[b[k] for k in spec.in_features], train=True)
upd = agent.PolicyUpdater(loss)
for _ in range(int(os.environ.get("GRL_STEPS", "3"))):
    upd.step(b)
torch.cuda.synchronize()
print("done")
