"""Rollout driver on the GPU: GAE from the HIP critic, env-aligned sampling without replacement, gather-assembled minibatches
(eager and graph-replayed) -- the final parameters must equal those of the same updates fed with explicitly indexed batches."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(N, T, seed):
    from geometry_rl_amd import agent, graph, synthetic as syn
    dev = torch.device("cuda:0")
    spec = graph.rigid_spec()
    cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
    torch.manual_seed(0)
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
    frames = []
    for t in range(T + 1):  # one synthetic frame set per time step (same env -> same point count: env_offset 0)
        b = dict(syn.make_rigid_obs(N, seed=seed + t))
        b.update(syn.make_ppo_fields(N, 6, seed=seed + t))
        frames.append(b)
    data = {k: torch.stack([f[k] for f in frames[:T]], dim=1).to(dev) for k in frames[0]}
    g = syn.make_gae_inputs(N, T, seed=seed)
    data.update(reward=g["reward"].reshape(N, T, 1).to(dev), done=g["done"].reshape(N, T, 1).to(dev),
                terminated=g["terminated"].reshape(N, T, 1).to(dev))
    next_last = {k: frames[T][k].unsqueeze(1).to(dev) for k in spec.in_features}
    with torch.no_grad():
        actor.forward_diag(*[data[k][:, 0].contiguous() for k in spec.in_features], train=True)   # calibration
    return spec, cfg, loss, data, next_last


@pytest.mark.parametrize("use_graph", [False, True])
def test_driver_matches_explicit_loop(use_graph):
    from geometry_rl_amd import agent
    from geometry_rl_amd.rollout import RolloutBuffer, RolloutDriver
    N, T = 8, 4
    # reference: same sampler, batches built with index_select, eager updates
    spec, cfg, loss, data, next_last = _make(N, T, seed=21)
    upd = agent.PolicyUpdater(loss, lr=cfg.lr)
    buf = RolloutBuffer(dict(data))
    drv = RolloutDriver(upd, spec, ppo_epochs=2, seed=5)
    drv.compute_advantages(buf, next_last)
    adv_ref = buf.data["advantage"].clone()
    keys = list(spec.in_features) + ["action", "loc", "var", "sample_log_prob", "state_value", "advantage", "value_target"]
    for idx in drv.minibatches(buf):
        upd.step(buf.rows(idx, keys))
    ref_flat = upd.flat.detach().cpu()

    spec, cfg, loss, data, next_last = _make(N, T, seed=21)
    upd2 = agent.PolicyUpdater(loss, lr=cfg.lr, use_graph=use_graph)
    buf2 = RolloutBuffer(dict(data))
    drv2 = RolloutDriver(upd2, spec, ppo_epochs=2, seed=5)
    out = drv2.run(buf2, next_last)
    assert torch.equal(buf2.data["advantage"], adv_ref)
    assert upd2.steps == 2 * T and out is not None
    err = (upd2.flat.detach().cpu() - ref_flat).abs().max().item()
    print("max |param diff|", err)
    assert err <= 1e-7


@pytest.mark.parametrize("form", ["unrolled", "cursor"])
def test_several_steps_per_launch_equal_the_step_by_step_loop(form):
    """PolicyUpdater.run_minibatches (round 6): ``unroll`` minibatch steps recorded into ONE graph per lane -- in-graph gathers from a static
    index matrix, private inputs per lane, the critic's gate as a launch of its lane, no join between the steps of a launch.  Ten minibatches
    per epoch with unroll = 4: an eager first step, two 4-step launches, one single step -- the parameters, both Adam moments and every
    step's loss dict must be those of the step-by-step loop."""
    from geometry_rl_amd import agent
    from geometry_rl_amd.rollout import RolloutBuffer, RolloutDriver
    N, T = 8, 10
    res = {}
    want_form = form
    for form in ("loop", "launches"):
        spec, cfg, loss, data, next_last = _make(N, T, seed=33)
        upd = agent.PolicyUpdater(loss, lr=cfg.lr, use_graph=True)
        upd.epoch_unroll = 4 if form == "launches" else 1
        if want_form == "cursor":   # the form of the gated sizes (one step per launch, index row by device cursor), forced at this toy size
            upd.epoch_unroll_max_gated_frames, upd.epoch_gated_from_frames, upd.epoch_cursor = 0, 0, True   # (off by default: measured no better than the per-step program)
        buf = RolloutBuffer(dict(data))
        drv = RolloutDriver(upd, spec, ppo_epochs=2, seed=9)
        drv.compute_advantages(buf, next_last)
        kls = []
        if form == "loop":
            for idx in drv.minibatches(buf):
                kls.append(upd.step_from(buf, idx)["kl"].clone())
            assert upd._epoch is None
        else:
            dev = next(iter(buf.data.values())).device
            for _ in range(2):
                idxs = drv.epoch_minibatches(buf.N, buf.T, dev)
                out = upd.run_minibatches(buf, torch.stack(idxs))
                assert upd._epoch is not None and upd._epoch["key"][2] == (want_form == "cursor")
                assert len(upd.last_outs) == (1 if want_form == "cursor" else 4)
            kls = ([o["kl"].clone() for o in upd.last_outs] if want_form == "unrolled" else []) + [out["kl"].clone()]
        torch.cuda.synchronize()
        assert upd.steps == 2 * T and int(upd.step_dev.item()) == 2 * T and int(upd.step_dev_c.item()) == 2 * T
        res[form] = (upd.flat.detach().clone(), upd.exp_avg.detach().clone(), upd.exp_avg_sq.detach().clone(), kls)
    for a, b in zip(res["loop"][:3], res["launches"][:3]):
        assert torch.equal(a, b), (a - b).abs().max().item()
    # epoch 2 of the launch form: steps 11-14, 15-18 by launches (last_outs = steps 15-18), 19-20 singly (out = step 20)
    loop_kls = res["loop"][3]
    for got, want in zip(res["launches"][3], (loop_kls[14:18] if want_form == "unrolled" else []) + [loop_kls[19]]):
        assert torch.equal(got, want)


def test_measured_choice_of_the_recorded_form_changes_nothing_but_the_time():
    """Above 64 work-frames run_minibatches MEASURES which recorded form the size takes (PolicyUpdater._tune_form: the multi-step launch and the
    per-step program in alternating blocks, HIP events, one synchronisation) as soon as a call brings enough minibatches.  The measurement is
    made of ordinary updates of the next minibatches, and both forms are bitwise the step-by-step loop -- so is the whole call, whichever
    form wins; a second call takes the measured form without measuring again."""
    from geometry_rl_amd import agent
    from geometry_rl_amd.rollout import RolloutBuffer, RolloutDriver
    N, T = 96, 30
    res = {}
    for form in ("loop", "measured"):
        spec, cfg, loss, data, next_last = _make(N, T, seed=57)
        upd = agent.PolicyUpdater(loss, lr=cfg.lr, use_graph=True)
        upd.epoch_unroll = 4
        buf = RolloutBuffer(dict(data))
        drv = RolloutDriver(upd, spec, ppo_epochs=1, seed=5)
        drv.compute_advantages(buf, next_last)
        dev = next(iter(buf.data.values())).device
        idx = torch.stack(drv.epoch_minibatches(buf.N, buf.T, dev))     # [30, 96]
        if form == "loop":
            upd.autotune_form = False
            for j in range(T):
                upd.step_from(buf, idx[j])
        else:
            assert upd.autotune_form and upd.tune_minibatches(4) == 23
            upd.run_minibatches(buf, idx[:25])           # eager first step, 23 measured steps, one single step
            assert upd.form_by_size.get(N) in ("unrolled", "per_step")
            times = dict(upd.form_times[N])
            assert times["unrolled_ms_per_step"] > 0 and times["per_step_ms_per_step"] > 0
            upd.run_minibatches(buf, idx[25:])           # five steps in the measured form
            assert upd.form_times[N] == times            # (not measured again)
        torch.cuda.synchronize()
        assert upd.steps == T and int(upd.step_dev.item()) == T and int(upd.step_dev_c.item()) == T
        res[form] = (upd.flat.detach().clone(), upd.exp_avg.detach().clone(), upd.exp_avg_sq.detach().clone())
    for a, b in zip(res["loop"], res["measured"]):
        assert torch.equal(a, b), (a - b).abs().max().item()


def test_launch_forms_can_be_mixed_and_follow_a_changed_hyper_parameter():
    """run_minibatches (eight-step launches), step_from (the per-step program) and a changed hyper-parameter in between: the step counts live on
    the device, every program reads the same flat buffers, and a changed Adam epsilon drops EVERY recording (the multi-step one included) -- the
    result must be the step-by-step loop's."""
    from geometry_rl_amd import agent
    from geometry_rl_amd.rollout import RolloutBuffer, RolloutDriver
    N, T = 8, 12
    res = {}
    for form in ("loop", "mixed"):
        spec, cfg, loss, data, next_last = _make(N, T, seed=41)
        upd = agent.PolicyUpdater(loss, lr=cfg.lr, use_graph=True)
        upd.epoch_unroll = 4
        buf = RolloutBuffer(dict(data))
        drv = RolloutDriver(upd, spec, ppo_epochs=1, seed=3)
        drv.compute_advantages(buf, next_last)
        dev = next(iter(buf.data.values())).device
        idx = torch.stack(drv.epoch_minibatches(buf.N, buf.T, dev))     # [12, 8]
        if form == "loop":
            for j in range(T):
                if j == 7:
                    upd.eps = 1e-4
                upd.step_from(buf, idx[j])
        else:
            upd.run_minibatches(buf, idx[:5])            # eager first step + one 4-step launch
            upd.step_from(buf, idx[5])                   # the per-step program (records itself)
            upd.step_from(buf, idx[6])
            assert upd._epoch is not None
            upd.eps = 1e-4                               # a recorded scalar changes: every recording is dropped
            assert upd._epoch is None and upd._program is None
            upd.run_minibatches(buf, idx[7:])            # five steps: one 4-step launch (recorded again) + one single step
        torch.cuda.synchronize()
        assert upd.steps == T and int(upd.step_dev.item()) == T and int(upd.step_dev_c.item()) == T
        res[form] = (upd.flat.detach().clone(), upd.exp_avg_sq.detach().clone())
    for a, b in zip(res["loop"], res["mixed"]):
        assert torch.equal(a, b), (a - b).abs().max().item()


def test_collect_normalise_update_round_trip():
    """One whole training iteration on the device with a synthetic environment: raw observations -> running normalisation + clip ->
    collector-side actor (sampling) -> rollout buffer -> critic values + shifted GAE -> minibatch updates.  Checks the plumbing (keys,
    shapes, finite losses, parameters move) -- every piece is checked numerically by its own test."""
    from geometry_rl_amd import agent, graph, synthetic as syn
    from geometry_rl_amd.rollout import PolicyActor, RolloutDriver, collect
    from geometry_rl_amd.transforms import ObservationNormalizer
    dev = torch.device("cuda:0")
    N, T = 8, 3
    spec = graph.rigid_spec()
    cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
    torch.manual_seed(0)
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
    raw_keys = [k for k in spec.in_features if not k.startswith("norm_")]
    state = {"t": 0}

    def raw_obs(t):
        o = syn.make_rigid_obs(N, seed=70 + t)
        return {k: o[k].to(dev) for k in raw_keys}

    def env_step(action):
        assert action.shape == (N, 6)
        state["t"] += 1
        g = torch.Generator().manual_seed(state["t"])
        return (raw_obs(state["t"]), torch.randn(N, generator=g).to(dev), (torch.rand(N, generator=g) < 0.2).to(dev),
                torch.zeros(N, dtype=torch.bool, device=dev))

    norm = ObservationNormalizer(device=dev)
    buf, next_last = collect(env_step, raw_obs(0), PolicyActor(actor, spec, use_graph=False), T, normalizer=norm)
    assert buf.N == N and buf.T == T and set(spec.in_features) <= set(buf.data)
    assert buf.data["action"].shape == (N, T, 6) and buf.data["sample_log_prob"].shape[:2] == (N, T)
    upd = agent.PolicyUpdater(loss, lr=cfg.lr)
    p0 = upd.flat.clone()
    out = RolloutDriver(upd, spec, ppo_epochs=1, seed=1).run(buf, next_last)
    assert all(torch.isfinite(out[k]).all() for k in ("loss_objective", "loss_trust_region", "loss_critic", "kl"))
    assert float((upd.flat - p0).abs().max()) > 0


@pytest.mark.parametrize("N,T,budget", [(8, 4, None), (96, 7, None), (96, 7, 3 * 96 * 33 * 256)])
def test_time_batched_critic_pass_is_the_loop_bitwise(N, T, budget):
    """The once-per-rollout critic pass (train.py:249-251): all time steps as groups of ONE launch set (per-step LayerNorm statistic
    slots) must give bitwise the values of the reference's loop over T (gnn_vf_net.py:72-80) -- also when the byte budget cuts the
    time axis into chunks (third case: three steps per chunk)."""
    spec, cfg, loss, data, next_last = _make(N, T, seed=33)
    critic = loss.critic_network
    vf = critic._network1
    obs = [data[k] for k in spec.in_features]
    with torch.no_grad():
        looped = torch.stack([vf._values([a[:, i].contiguous() for a in obs], False) for i in range(T)], dim=1)
        if budget is not None:
            vf.GROUPED_BYTES = budget
        batched = critic(*obs, train=False)
    assert batched.shape == (N, T, 1)
    assert torch.equal(batched.reshape(N, T), looped), float((batched.reshape(N, T) - looped).abs().max())
    # with autograd on, the 3-D call is still the differentiable loop
    v = critic(*[a[:, :2] for a in obs], train=True)
    assert v.requires_grad and v.shape == (N, 2, 1)
