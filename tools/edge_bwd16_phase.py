"""Phase timing of the fused edge backward (diagnostic build -DGRL_B16_PHASE, GRL_LIB=<that .so>): shares of wave 0's s_memtime ticks per
stage of a pass, summed over every workgroup, for one 4096-frame policy update of the bench workload."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from geometry_rl_amd import agent, graph, hip, synthetic as syn
dev = torch.device("cuda:0")
WL = os.environ.get("GRL_WORKLOAD", "rigid_hepi")   # any bench workload; rope_hepi_bf16 reads the bf16 twin's counters
spec, cfg, make_obs, _ = bench.workload(WL)
actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
B = int(os.environ.get("GRL_B", "4096"))
b = dict(make_obs(B, 1, 0)); b.update(syn.make_ppo_fields(B, spec.num_actuators * cfg.output_dim_vec * 3, seed=1))
b = {k: v.to(dev) for k, v in b.items()}
upd = agent.PolicyUpdater(loss, lr=cfg.lr, clip_grad_norm=cfg.clip_grad_norm, max_grad_norm=cfg.max_grad_norm)
READ = getattr(hip.lib(), "grl_edge_bwd16_phase_read" + ("_bf16" if cfg.precision == "bf16" else ""))
buf = (ctypes.c_ulonglong * 16)()
for _ in range(2):
    upd.step(b)
torch.cuda.synchronize()
READ(buf, ctypes.c_int(1))
upd.step(b); torch.cuda.synchronize()
READ(buf, ctypes.c_int(1))
names = ["pass top, rest: invariants, phi, split", "layer 1: 12 MFMA, GELU + derivative, split", "layer 2: 24 MFMA, GELU + derivative, split",
         "K: 24 MFMA, d x_src, dK, staging dK | g2, transposed reads", "dZ2: 24 MFMA, dWk: 12 MFMA 32x32, split", "staging dZ2 | g1, dZ1: 24 MFMA, dW2: 12 MFMA 32x32, split",
         "staging dZ1 | phi, dW1: 6 MFMA 32x32",
         "pass top: loop tail of the previous pass", "pass top: prefetched dM row taken over (vmcnt wait)",
         "pass top: layer-1 fragments + transposed dW1 operands requested (stamp waits for the LDS reads)",
         "pass top: node change (flush, skip of empty nodes, node_begin), next gather issued"]
v = [buf[i] for i in range(11)]
tot = sum(v) or 1
print(f"== edge_bwd16_kernel, workload {WL}: {tot / 1e6:.1f} Mticks over wave 0 of every workgroup")
for n, x in zip(names, v):
    print(f"   {n:64s} {100 * x / tot:5.1f} %")
