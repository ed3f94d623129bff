"""BASELINE config 1: rigid_insertion_multi_transformer_trpl, 64 synthetic envs x 32 steps (plumbing case).

The transformer baseline actor is stock torch (pinned to the reference module on the CPU: tests/test_transformer_fixture.py); what this
test covers is that it runs on the SAME machinery as the HEPi path -- observation features by one HIP launch, post_fc Gaussian head,
fused TRPL kernel, HIP DeepSets critic, flat-buffer Adam, recorded step, rollout driver -- and lands where a plain CPU pipeline lands:
the same torch modules on the CPU + the oracle's critic, TRPL loss and two torch.optim.Adam(eps=1e-5)."""
import copy

import pytest
import torch
import torch.nn.functional as F

from oracle import graph as ogr, step as ost, trpl as otr
from geometry_rl_amd import synthetic as syn

pytestmark = pytest.mark.gpu
LOSS_KEYS = ["loss_objective", "loss_trust_region", "loss_entropy", "loss_critic", "ESS", "kl", "constraint", "mean_constraint",
             "mean_constraint_max", "cov_constraint", "cov_constraint_max", "entropy", "entropy_diff"]


def _agent(dev):
    from geometry_rl_amd import agent, graph
    spec = graph.rigid_spec()
    cfg = agent.AgentConfig(model="transformer", output_dim=2, output_dim_vec=2)   # A = 6
    torch.manual_seed(3)
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
    with torch.no_grad():   # the orthogonal(0.01) heads would leave loc ~ 0: make the comparison see the transformer
        actor._mean.weight.mul_(30.0)
        actor._pre_std.weight.mul_(30.0)
    return spec, cfg, actor, critic, loss


def test_one_update_matches_cpu_pipeline():
    from geometry_rl_amd import agent
    from test_gpu_step import check
    dev = torch.device("cuda:0")
    B = 64
    spec, cfg, actor, critic, loss = _agent(dev)
    assert loss.in_features == ["scalars", "norm_position_vectors", "norm_velocity_vectors", "norm_position_vectors",
                                "norm_velocity_vectors", "infos"]          # configs/rigid_insertion_multi_transformer_trpl_cfg.yaml:88-94
    batch = dict(syn.make_rigid_obs(B, seed=31))
    batch.update(syn.make_ppo_fields(B, 6, seed=31))
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    # ---- CPU pipeline: the same torch modules + the oracle's graph features, critic, loss
    o_spec = ogr.rigid_spec()
    o_cfg = ost.AgentConfig()
    c_par = {k[len("_network1."):]: v.detach().cpu().clone() for k, v in critic.state_dict().items()}
    oracle = ost.OracleAgent(o_spec, o_cfg, {"_unused": torch.zeros(1)}, c_par)   # only its graph features / critic / optimizer are used
    gnn_c, mean_c, std_c = copy.deepcopy(actor.gnn).cpu(), copy.deepcopy(actor._mean).cpu(), copy.deepcopy(actor._pre_std).cpu()
    a_params = list(gnn_c.parameters()) + list(mean_c.parameters()) + list(std_c.parameters())
    a_opt = torch.optim.Adam(a_params, lr=cfg.lr, eps=1e-5)
    a_obs = {"scalars": batch["scalars"], "position_vectors": batch["norm_position_vectors"],
             "velocity_vectors": batch["norm_velocity_vectors"], "norm_position_vectors": batch["norm_position_vectors"],
             "norm_velocity_vectors": batch["norm_velocity_vectors"], "infos": batch["infos"]}
    topo, graph, s, v = oracle._graph(a_obs, full_graph_obs=False, dist_as_pos=True)
    x = ogr.critic_input(topo, s, v)

    class G:
        batch_size, node_types, output_mask_key = B, topo["node_types"], "grippers"
        nodes_per_sample = topo["n_per"]

    hidden = gnn_c.one_step(G(), x)
    loc = mean_c(hidden).reshape(B, -1)
    std = F.softplus(std_c(hidden) + otr.inverse_softplus(torch.tensor(1.0 - 1e-5))) + 1e-5
    var = std.reshape(B, -1) ** 2
    value = oracle.critic_forward({k: batch[k] for k in o_spec.in_features})
    ref = otr.trpl_loss(loc, var, batch, value, mean_bound=cfg.mean_bound, cov_bound=cfg.cov_bound,
                        trust_region_coeff=cfg.trust_region_coeff, entropy_coef=cfg.entropy_coef, critic_coef=cfg.critic_coef,
                        clip_value=cfg.clip_value, proj_type="kl")
    (ref["loss_objective"] + ref["loss_entropy"] + ref["loss_trust_region"]).backward()
    ref["loss_critic"].backward()
    a_opt.step()
    oracle.critic_optim.step()
    # ---- the package: one recorded-step-capable update
    upd = agent.PolicyUpdater(loss, lr=cfg.lr)
    out = upd.step(dbatch)
    check("loc", out["loc"], loc)
    check("var", out["sigma"] ** 2, var)
    check("state_value", out["state_value"], value)
    for k in LOSS_KEYS:
        check(k, out[k], ref[k])
    for (k, p), q in zip(list(actor.gnn.named_parameters()), gnn_c.parameters()):
        check("param gnn." + k, p, q, 2e-5)
    check("param _mean.weight", actor._mean.weight, mean_c.weight, 2e-5)
    check("param _pre_std.weight", actor._pre_std.weight, std_c.weight, 2e-5)
    for k, p in critic.named_parameters():
        check("param " + k, p, oracle.critic[k[len("_network1."):]], 2e-5)


@pytest.mark.parametrize("use_graph", [False, True])
def test_config1_rollout_pass_64_envs_x_32_steps(use_graph):
    """GAE over the 64 x 32 rollout, then 5 epochs x 32 minibatches of 64 frames through the rollout driver (eager and recorded)."""
    from geometry_rl_amd import agent
    from geometry_rl_amd.rollout import RolloutBuffer, RolloutDriver
    dev = torch.device("cuda:0")
    N, T = 64, 32
    spec, cfg, actor, critic, loss = _agent(dev)
    frames = []
    for t in range(T):
        b = dict(syn.make_rigid_obs(N, seed=200 + t))
        b.update(syn.make_ppo_fields(N, 6, seed=300 + t))
        frames.append({k: v.to(dev) for k, v in b.items()})
    data = {k: torch.stack([f[k] for f in frames], dim=1) for k in frames[0]}
    g = syn.make_gae_inputs(N, T, seed=5, episode_len=16)
    data.update(reward=g["reward"].reshape(N, T, 1).to(dev), done=g["done"].reshape(N, T, 1).to(dev),
                terminated=g["terminated"].reshape(N, T, 1).to(dev))
    buf = RolloutBuffer(data)
    upd = agent.PolicyUpdater(loss, lr=cfg.lr, use_graph=use_graph)
    assert upd.mode.startswith("eager")   # asking for a recorded step with the stock-torch actor degrades loudly (mode says so), not silently
    drv = RolloutDriver(upd, spec, ppo_epochs=5, seed=0)
    p0 = upd.flat.clone()
    out = drv.run(buf, {k: frames[0][k].unsqueeze(1) for k in spec.in_features})
    assert upd.steps == 5 * T
    for k in LOSS_KEYS:
        assert torch.isfinite(out[k]).all(), k
    assert float((upd.flat - p0).abs().max()) > 0
    test_config1_rollout_pass_64_envs_x_32_steps.results = getattr(test_config1_rollout_pass_64_envs_x_32_steps, "results", {})
    test_config1_rollout_pass_64_envs_x_32_steps.results[use_graph] = upd.flat.detach().cpu().clone()
    r = test_config1_rollout_pass_64_envs_x_32_steps.results
    if len(r) == 2:   # recorded replay == eager launches after 160 updates
        assert (r[True] - r[False]).abs().max() <= 5e-5
