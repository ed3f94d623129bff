"""A few policy-update steps of a bench workload, for rocprofv3 (kernel trace or PMC passes).
   GRL_WORKLOAD = rigid_hepi (default) | cloth_hepi | rope_hepi | rope_hepi_var | rope_hepi_bf16 | rigid2_empn;  GRL_B frames;  GRL_STEPS"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from geometry_rl_amd import agent, synthetic as syn
dev = torch.device("cuda:0")
spec, cfg, make_obs, _ = bench.workload(os.environ.get("GRL_WORKLOAD", "rigid_hepi"))
actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
B = int(os.environ.get("GRL_B", "4096"))
b = dict(make_obs(B, 1, 0)); b.update(syn.make_ppo_fields(B, spec.num_actuators * cfg.output_dim_vec * 3, seed=1))
b = {k: v.to(dev) for k, v in b.items()}
with torch.no_grad():
    actor.forward_diag(*[b[k] for k in spec.in_features], train=True)
upd = agent.PolicyUpdater(loss, lr=cfg.lr, clip_grad_norm=cfg.clip_grad_norm, max_grad_norm=cfg.max_grad_norm)
for _ in range(int(os.environ.get("GRL_STEPS", "3"))):
    upd.step(b)
torch.cuda.synchronize()
print("done")
