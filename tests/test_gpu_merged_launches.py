"""Merged launches of round 6 (csrc/node_ops.hip step_head_kernel / lift_fiber_basis_bwd_kernel, lane signals riding on the fiber convolution
and on the tail; include/grl_hip.h "merged launches"): roles that do not depend on each other share one launch, told apart by block range, and
run the stand-alone kernels' own device functions -- so a recorded policy update with the merges ON must land BITWISE on the one with every role
as its own launch (parameters, both Adam moments, every reported loss value), on one rank with the two-lane program, for the HEPi and the EMPN
actor, in the fp32 and the bf16 build."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(model, precision, merged, steps=3):
    from geometry_rl_amd import agent, graph, ops, synthetic as syn
    dev = torch.device("cuda:0")
    old = (ops.FUSE_HEAD, ops.FUSE_TAIL_PRE, ops.SIGNAL_IN_KERNEL)
    ops.FUSE_HEAD = ops.FUSE_TAIL_PRE = ops.SIGNAL_IN_KERNEL = merged
    try:
        if model == "empn":
            spec = graph.rigid_spec(G=2, angular_velocity=False, object_velocity=False)
            cfg = agent.AgentConfig(model="empn", precision=precision)
            obs = syn.make_rigid_obs(24, G=2, angular_velocity=False, object_velocity=False, seed=8)
        else:
            spec = graph.rigid_spec()
            cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2, precision=precision)
            obs = syn.make_rigid_obs(24, seed=3)
        torch.manual_seed(0)
        actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
        A = spec.num_actuators * cfg.output_dim_vec * 3
        batch = dict(obs)
        batch.update(syn.make_ppo_fields(24, A, seed=5))
        batch = {k: v.to(dev) for k, v in batch.items()}
        with torch.no_grad():
            actor.forward_diag(*[batch[k] for k in spec.in_features], train=True)    # calibration
        upd = agent.PolicyUpdater(loss, lr=cfg.lr, use_graph=True)
        outs = []
        for _ in range(steps):   # eager, recording, replay
            o = upd.step(batch)
            outs.append({k: v.detach().clone() for k, v in o.items() if torch.is_tensor(v) and v.numel() == 1})
        torch.cuda.synchronize()
        assert upd.mode.startswith("graph")
        return upd.flat.clone(), upd.exp_avg.clone(), upd.exp_avg_sq.clone(), outs
    finally:
        ops.FUSE_HEAD, ops.FUSE_TAIL_PRE, ops.SIGNAL_IN_KERNEL = old


@pytest.mark.parametrize("model,precision", [("hepi", "fp32"), ("hepi", "bf16"), ("empn", "fp32")])
def test_merged_launches_equal_separate_launches_bitwise(model, precision):
    a = _run(model, precision, merged=False)
    b = _run(model, precision, merged=True)
    for name, u, v in zip(("parameters", "exp_avg", "exp_avg_sq"), a[:3], b[:3]):
        assert torch.equal(u, v), (name, (u - v).abs().max().item())
    for step, (oa, ob) in enumerate(zip(a[3], b[3])):
        assert set(oa) == set(ob)
        for k in oa:
            assert torch.equal(oa[k], ob[k]), (step, k, float(oa[k]), float(ob[k]))
