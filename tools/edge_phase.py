"""Phase timing of the three edge kernels (diagnostic build with -DGRL_PHASE_PROF, GRL_LIB=<that .so>): shares of one wave's cycles per
stage, accumulated over every workgroup's wave 1, for one 4096-frame policy update of the bench workload."""
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
buf = (ctypes.c_ulonglong * 72)()
for _ in range(2):
    upd.step(b)
torch.cuda.synchronize()
hip.lib().grl_edge_phase_read(buf, ctypes.c_int(1))
upd.step(b); torch.cuda.synchronize()
hip.lib().grl_edge_phase_read(buf, ctypes.c_int(1))
names = {0: "tile head (rowptr, first indices, positions)", 1: "pass top (next indices, row loads issued)", 2: "(a,b) ready + poly + split",
         3: "layer 1 (2 tiles: frags, 3 MFMA, GELU)", 4: "split g1", 5: "layer 2 (2 tiles: frags, 12 MFMA, GELU)", 6: "split g2",
         7: "kernel layer (2 tiles: frags, 12 MFMA, message)", 8: "next positions issued + invariants", 9: "tile tail (fold, stores)",
         10: "w: dK, split, dWk (4 transposes, 24 MFMA)", 11: "w: dZ2 (24 MFMA, * gelu')", 12: "w: split dZ2, dW2 (4 transposes, 24 MFMA)",
         13: "w: dZ1 (24 MFMA)", 14: "w: split dZ1, dW1 (3 transposes, 12 MFMA)"}
for k, kname in enumerate(("edge_conv_fwd_kernel", "edge_conv_bwd_x_kernel", "edge_conv_bwd_w_kernel")):
    v = [buf[k * 24 + i] for i in range(24)]
    tot = sum(v) or 1
    print(f"== {kname}: {tot / 1e6:.1f} Mticks over the sampled waves")
    for i in range(24):
        if v[i]:
            print(f"   {names.get(i, str(i)):52s} {100 * v[i] / tot:5.1f} %")
