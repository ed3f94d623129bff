"""The tensordict / torchrl branches of geometry_rl_amd.trpl (``_TensorDict``, ``_LossBase = torchrl.objectives.LossModule``) never run
on the build or GPU images: neither package is installed.  This test puts minimal stand-ins (tests/stubs/: our own, not upstream code)
on the path of a FRESH interpreter so that those branches execute at least once: class creation on the torchrl base, construction with
the reference keyword arguments, batch extraction from a TensorDict, and -- on a GPU box -- one forward that returns a TensorDict."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, timeout=600):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests", "stubs"), ROOT, env.get("PYTHONPATH", "")])
    p = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return p.stdout


def test_class_and_batch_extraction_on_stub_packages():
    out = _run("""
        import torch, tensordict, torchrl.objectives as tro
        from geometry_rl_amd import trpl
        assert trpl._TensorDict is tensordict.TensorDict and trpl._LossBase is tro.LossModule
        assert issubclass(trpl.TRPLLoss, tro.LossModule)
        class Spec: in_features = ["a", "b"]
        class HD: spec = Spec()
        class Actor(torch.nn.Module):
            hyper_data = HD()
            def forward_diag(self, *a, **k): raise NotImplementedError
        class Critic(torch.nn.Module):
            _network1 = None
        loss = trpl.TRPLLoss(Actor(), Critic(), projection=trpl.KLProjectionLayer(mean_bound=0.05, cov_bound=0.0005), entropy_coef=0.005,
                             critic_coef=0.5, trust_region_coef=8.0, clip_value=0.2, normalize_advantage=True)
        assert loss.in_features == ["a", "b"] and "loss_critic" in loss.out_keys
        td = tensordict.TensorDict({"a": torch.ones(3, 2), "b": torch.zeros(3, 1), "action": torch.zeros(3, 6), "loc": torch.zeros(3, 6),
                                    "var": torch.ones(3, 6), "sample_log_prob": torch.zeros(3), "advantage": torch.zeros(3, 1),
                                    "value_target": torch.zeros(3, 1), "state_value": torch.zeros(3, 1)}, [3])
        b = trpl._as_batch(td, loss.in_features)
        assert set(["a", "b", "action", "loc", "var", "advantage"]) <= set(b)
        print("ok")
    """)
    assert "ok" in out


@pytest.mark.gpu
def test_forward_returns_tensordict_with_torchrl_base():
    out = _run("""
        import torch, tensordict
        from geometry_rl_amd import agent, graph, synthetic as syn, trpl
        dev = torch.device("cuda:0")
        spec = graph.rigid_spec()
        cfg = agent.AgentConfig(only_upper_hemisphere=True, output_dim=2, output_dim_vec=2)
        torch.manual_seed(0)
        actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
        import torchrl.objectives as tro
        assert isinstance(loss, tro.LossModule)
        B = 8
        batch = dict(syn.make_rigid_obs(B, seed=3)); batch.update(syn.make_ppo_fields(B, 6, seed=3))
        td = tensordict.TensorDict({k: v.to(dev) for k, v in batch.items()}, [B])
        out = loss(td)
        assert isinstance(out, tensordict.TensorDict)
        sel = out.select("loss_objective", "loss_trust_region", "loss_entropy", "loss_critic").detach()
        vals = {k: float(v) for k, v in sel.items()}
        assert all(v == v for v in vals.values()), vals
        (out["loss_objective"] + out["loss_entropy"] + out["loss_trust_region"]).backward()
        out["loss_critic"].backward()
        g = sum(float(p.grad.abs().sum()) for p in actor.parameters() if p.grad is not None)
        assert g > 0
        print("ok", vals)
    """)
    assert "ok" in out
