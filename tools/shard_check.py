"""Debug: are per-frame outputs independent of how the minibatch is sharded?  And is a launch run-to-run deterministic?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import agent, graph, synthetic as syn
dev = torch.device("cuda:0")
spec = graph.rigid_spec(G=2, angular_velocity=False, object_velocity=False)
cfg = agent.AgentConfig()
torch.manual_seed(0)
actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
B = 16
b = dict(syn.make_rigid_obs(B, G=2, angular_velocity=False, object_velocity=False, seed=4)); b.update(syn.make_ppo_fields(B, 6, seed=4))
b = {k: v.to(dev) for k, v in b.items()}
obs = [b[k] for k in spec.in_features]
with torch.no_grad():
    actor.forward_diag(*obs, train=True)
    l1, s1 = actor.forward_diag(*obs)
    l2, s2 = actor.forward_diag(*obs)
    print("rerun diff", (l1 - l2).abs().max().item(), (s1 - s2).abs().max().item())
    la, sa = actor.forward_diag(*[o[:8].contiguous() for o in obs])
    lb, sb = actor.forward_diag(*[o[8:].contiguous() for o in obs])
    print("shard diff loc", (torch.cat([la, lb]) - l1).abs().max().item(), "sigma", (torch.cat([sa, sb]) - s1).abs().max().item())
    v1 = critic(*obs)
    print("value rerun", (critic(*obs) - v1).abs().max().item())
def grads(batch):
    for p in list(actor.parameters()) + list(critic.parameters()):
        p.grad = None
    out = loss(batch)
    (out["loss_objective"] + out["loss_entropy"] + out["loss_trust_region"]).backward()
    out["loss_critic"].backward()
    return {k: p.grad.clone() for k, p in actor.named_parameters() if p.grad is not None}
g1 = grads(b); g2 = grads(b)
worst = max(((g1[k] - g2[k]).abs().max().item(), k) for k in g1)
print("grad rerun worst", worst)
for k in list(g1)[:40]:
    d = (g1[k] - g2[k]).abs().max().item()
    if d > 0: print("  ", k, d, g1[k].abs().max().item())
