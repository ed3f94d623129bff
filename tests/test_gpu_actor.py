"""Collector-side actor: no-grad policy pass + MultivariateNormal sampling with log-prob, eager and replayed from a
hipGraph, against torch.distributions (what the reference's ProbabilisticActor calls, configs/algorithm/policy/default.yaml:6)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_policy_actor_matches_torch_distribution():
    from geometry_rl_amd import agent, graph, synthetic as syn
    from geometry_rl_amd.rollout import PolicyActor
    dev = torch.device("cuda:0")
    spec = graph.rigid_spec(P=8, G=1, E_mesh=4)
    cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
    torch.manual_seed(0)
    actor, _, _, _ = agent.build_agent(spec, cfg, device=dev)
    B = 64
    act = PolicyActor(actor, spec, use_graph=True, seed=5)
    seen = []
    for step in range(4):   # call 1 eager (calibration + topology), call 2 records, calls 3-4 replay
        obs = {k: v.to(dev) for k, v in syn.make_rigid_obs(B, P=8, G=1, E_mesh=4, seed=40 + step).items()}
        out = {k: v.clone() for k, v in act(obs).items()}
        loc, sigma = actor.forward_diag(*[obs[k] for k in spec.in_features], train=False)
        assert torch.equal(out["loc"], loc.detach()) or (out["loc"] - loc).abs().max() <= 1e-6
        assert (out["var"] - sigma.detach() ** 2).abs().max() <= 1e-7
        d = torch.distributions.MultivariateNormal(out["loc"].double().cpu(), covariance_matrix=out["var"].double().cpu().diag_embed())
        ref_lp = d.log_prob(out["action"].double().cpu())
        assert (out["sample_log_prob"].double().cpu() - ref_lp).abs().max() <= 1e-4, step
        z = (out["action"] - out["loc"]) / out["var"].sqrt()
        assert abs(float(z.mean())) < 0.25 and 0.7 < float(z.std()) < 1.3   # standard-normal draws
        seen.append(out["action"])
    assert not torch.equal(seen[2], seen[3])   # replays draw fresh noise
    # deterministic mode = the distribution's mode
    det = PolicyActor(actor, spec, use_graph=False, deterministic=True)
    out = det(obs)
    assert torch.equal(out["action"], out["loc"])
