"""Phase timing of edge_conv_bwd_w_kernel (debug build with -DGRL_W_PHASE_PROF, GRL_LIB=<that .so>)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import agent, graph, hip, ops, synthetic as syn
dev = torch.device("cuda:0")
spec = graph.rigid_spec()
cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
B = int(os.environ.get("GRL_B", "4096"))
b = dict(syn.make_rigid_obs(B, seed=1)); b.update(syn.make_ppo_fields(B, 6, seed=1))
b = {k: v.to(dev) for k, v in b.items()}
upd = agent.PolicyUpdater(loss)
buf = (ctypes.c_ulonglong * 16)()
for _ in range(2):
    upd.step(b)
torch.cuda.synchronize()
hip.lib().grl_w_phase_read(buf, ctypes.c_int(1))
upd.step(b); torch.cuda.synchronize()
hip.lib().grl_w_phase_read(buf, ctypes.c_int(1))
names = ["loop top (cur=nxt)", "indices+row loads issue", "chain (L1,L2,gelu both)", "invariants nxt + dK", "dWk (transposes+tn)", "dz2 (WkT product, *gp2)",
         "dW2 (+split dz2)", "dz1 (W2T product, *gp1)", "dW1"]
tot = sum(buf[i] for i in range(9))
for i in range(9):
    print(f"{names[i]:28s} {100 * buf[i] / tot:5.1f} %   {buf[i]/1e6:8.2f} Mticks")
