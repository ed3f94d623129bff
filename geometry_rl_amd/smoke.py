"""One tiny policy-update step on cuda:0 through the HIP kernels, checked against the CPU oracle (driver smoke test)."""
import torch


def run():
    from oracle import graph as ogr, step as ost
    from . import agent, graph, synthetic as syn
    dev = torch.device("cuda:0")
    B = 16
    kw = dict(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
    o_spec, spec = ogr.rigid_spec(), graph.rigid_spec()
    a_par, c_par = ost.init_agent_params(o_spec, ost.AgentConfig(**kw), seed=5)
    oracle = ost.OracleAgent(o_spec, ost.AgentConfig(**kw), a_par, c_par)
    actor, critic, proj, loss = agent.build_agent(spec, agent.AgentConfig(**kw), device=dev)
    actor.load_state_dict({k: v.to(dev) for k, v in a_par.items()}, strict=False)
    critic.load_state_dict({"_network1." + k: v.to(dev) for k, v in c_par.items()}, strict=False)
    batch = dict(syn.make_rigid_obs(B, seed=2))
    batch.update(syn.make_ppo_fields(B, 6, seed=2))
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    with torch.no_grad():  # first training call = data-dependent calibration of the conv kernels (conv.py:104-105)
        oracle.actor_forward({k: batch[k] for k in o_spec.in_features}, calibrate=True)
        actor.forward_diag(*[dbatch[k] for k in spec.in_features], train=True)
    for k, v in oracle.actor.items():  # std ratios of split-bf16 activations: 1e-4 (same bar as tests/test_gpu_step.py)
        if "kernel.weight" in k:
            err = (actor.state_dict()[k].cpu() - v).abs().max().item()
            assert err <= 1e-4 * max(1.0, v.abs().max().item()), ("calibration", k, err)
    # continue from identical (oracle-calibrated) weights so that the update below is compared on its own
    actor.load_state_dict({k: v.detach().to(dev) for k, v in oracle.actor.items()}, strict=False)
    actor._calib_checked = True
    upd = agent.PolicyUpdater(loss)
    out = upd.step(dbatch)
    ref, _ = oracle.update(batch)
    worst = 0.0
    for k in ("loc", "state_value", "loss_objective", "loss_trust_region", "loss_entropy", "loss_critic", "kl"):
        err = (out[k].detach().cpu().double() - ref[k].double()).abs().max().item()
        worst = max(worst, err)
        assert err <= 1e-4 * max(1.0, ref[k].abs().max().item()), (k, err)
    for k, p in actor.named_parameters():
        err = (p.detach().cpu() - oracle.actor[k]).abs().max().item()
        assert err <= 2e-5, (k, err)
    print(f"smoke ok: one HIP policy update matches the CPU oracle (worst abs err {worst:.2e})")
