"""Phase timing of the fused edge backward (diagnostic build -DGRL_B16_PHASE, GRL_LIB=<that .so>): shares of wave 0's s_memtime ticks per
stage of a pass, summed over every workgroup, for one 4096-frame policy update of the bench workload."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import agent, graph, hip, synthetic as syn
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
hip.lib().grl_edge_bwd16_phase_read(buf, ctypes.c_int(1))
upd.step(b); torch.cuda.synchronize()
hip.lib().grl_edge_bwd16_phase_read(buf, ctypes.c_int(1))
names = ["pass top: node change, gathers issued, invariants, phi", "layer 1: 12 MFMA, GELU + derivative, split", "layer 2: 24 MFMA, GELU + derivative, split",
         "K: 24 MFMA, d x_src, dK, staging dK | g2, transposed reads", "dZ2: 24 MFMA, dWk: 12 MFMA 32x32, split", "staging dZ2 | g1, dZ1: 24 MFMA, dW2: 12 MFMA 32x32, split",
         "staging dZ1 | phi, dW1: 6 MFMA 32x32"]
v = [buf[i] for i in range(7)]
tot = sum(v) or 1
print(f"== edge_bwd16_kernel: {tot / 1e6:.1f} Mticks over wave 0 of every workgroup")
for n, x in zip(names, v):
    print(f"   {n:64s} {100 * x / tot:5.1f} %")
