import os, sys, torch
sys.path.insert(0, "/root/repo")
from geometry_rl_amd import agent, graph, hip, ops, synthetic as syn
dev = torch.device("cuda:0")
spec = graph.rigid_spec()
cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
B = 512
b = dict(syn.make_rigid_obs(B, seed=1)); b.update(syn.make_ppo_fields(B, 6, seed=1))
b = {k: v.to(dev) for k, v in b.items()}
orig = ops.flush_deferred_grads
def dbg():
    jobs = ops.DEFERRED or []
    seen = {}
    for j in jobs:
        k = (j[0].data_ptr(), tuple(j[0].shape))
        seen.setdefault(k, 0); seen[k] += j[2]
    tot = 0
    for (ptr, shp), cols in seen.items():
        print("slab", shp, "cols used", cols, "MB", shp[0]*shp[1]*4/1e6); tot += shp[0]*cols*4
    print("n jobs", len(jobs), "total MB read", tot/1e6)
    orig()
ops.flush_deferred_grads = dbg
upd = agent.PolicyUpdater(loss)
upd.step(b); 
