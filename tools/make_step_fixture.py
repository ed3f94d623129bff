"""Fixture of ONE full policy update at a tiny shape ("Python harness rows"): inputs (observation groups, PPO fields,
initial parameters), every loss-dict entry, pre-clip gradients and post-Adam parameters, produced by the CPU oracle -- whose pieces
are pinned against the reference by tier1 / tier2 / tier2b / tier2c.  Guards the oracle against regressions (tests, CPU) and gives
the HIP path a fixed target that does not depend on running the oracle (tests, GPU)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geometry_rl_amd import synthetic as syn  # noqa: E402
from oracle import graph as ogr, step as ost  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    torch.set_num_threads(4)
    B = 6
    spec = ogr.rigid_spec(P=8, G=2, E_mesh=4, angular_velocity=False, object_velocity=False)
    cfg = ost.AgentConfig(clip_grad_norm=True, max_grad_norm=0.5)
    a, c = ost.init_agent_params(spec, cfg, seed=21)
    ag = ost.OracleAgent(spec, cfg, a, c)
    batch = dict(syn.make_rigid_obs(B, P=8, G=2, E_mesh=4, angular_velocity=False, object_velocity=False, seed=31))
    batch.update(syn.make_ppo_fields(B, spec.num_actuators * cfg.output_dim_vec * 3, seed=32))
    with torch.no_grad():
        ag.actor_forward({k: batch[k] for k in spec.in_features}, calibrate=True)
    rec = {"in." + k: v for k, v in batch.items()}
    rec.update({"actor0." + k: v.detach().clone() for k, v in ag.actor.items()})
    rec.update({"critic0." + k: v.detach().clone() for k, v in ag.critic.items()})
    out, grads = ag.update(batch)
    for k, v in out.items():
        if torch.is_tensor(v):
            rec["out." + k] = v
    rec.update({"grad.actor." + k: v for k, v in grads["actor"].items()})
    rec.update({"grad.critic." + k: v for k, v in grads["critic"].items()})
    rec.update({"actor1." + k: v.detach().clone() for k, v in ag.actor.items()})
    rec.update({"critic1." + k: v.detach().clone() for k, v in ag.critic.items()})
    np.savez(os.path.join(OUT, "step_rigid_g2_tiny.npz"), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in rec.items()})
    print("wrote", len(rec), "arrays")


if __name__ == "__main__":
    main()
