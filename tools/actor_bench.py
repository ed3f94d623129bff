"""Collector-side policy pass (no-grad forward + sampling) at the headline size: env steps per second of the policy alone."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometry_rl_amd import agent, graph, synthetic as syn
from geometry_rl_amd.rollout import PolicyActor
dev = torch.device("cuda:0")
spec = graph.rigid_spec()
cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
actor, _, _, _ = agent.build_agent(spec, cfg, device=dev)
B = 4096
obs = {k: v.to(dev) for k, v in syn.make_rigid_obs(B, seed=1).items()}
for graph_mode in (False, True):
    act = PolicyActor(actor, spec, use_graph=graph_mode)
    for _ in range(5):
        act(obs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        act(obs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"graph={graph_mode}: {dt*1e3:.3f} ms per policy pass of {B} envs = {B/dt/1e6:.2f} M env-steps/s")
