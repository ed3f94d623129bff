"""TRPL objective on the fused HIP kernel -- drop-in for
``geometry_rl/algorithms/trust_region_projections/objectives/trpl.py`` (TRPLLoss) and
``.../projections/kl_projection_layer.py`` (KLProjectionLayer, diagonal / contextual case).

The reference moves (mean, covariance) to the CPU, calls ITPAL per sample and comes back (trpl.py:241-245).  Here the
projection, log-ratio objective, trust-region regression, entropy bonus, clipped value loss, all metrics AND their
analytic gradients come out of one launch of ``grl_trpl_fwd_bwd``; nothing leaves the device and there is no host sync."""
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import hip, ops


class KLProjectionLayer:
    """Hyper-parameter holder with the reference constructor (base_projection_layer.py:123-197; configs/algorithm/projection/kl.yaml).
    ``__call__(policy, p, q, step)`` returns the projected (mean, "std") like the reference layer, for inspection."""

    def __init__(self, proj_type="kl", mean_bound=0.0, cov_bound=0.0, trust_region_coeff=0.0, scale_prec=True, mean_eq=False,
                 entropy_schedule=None, action_dim=None, total_train_steps=None, target_entropy=0.0, temperature=0.0,
                 entropy_eq=False, entropy_first=False, cpu=False, dtype=torch.float32, **ignored):
        if entropy_schedule:
            raise NotImplementedError("entropy_schedule is False in every reference TRPL config (kl.yaml:9)")
        kinds = {"kl": 0, "frob": 1, "frobenius": 1, "w2": 2, "wasserstein": 2}
        if proj_type.lower() not in kinds or mean_eq or not scale_prec:
            raise NotImplementedError("projections: kl | frob | w2 (commutative), each with the Mahalanobis (scale_prec) mean bound")
        self.proj_type, self.mean_bound, self.cov_bound = proj_type, float(mean_bound), float(cov_bound)
        self.proj_code = kinds[proj_type.lower()]
        self.trust_region_coeff = float(trust_region_coeff)
        self.initial_entropy = None

    def __call__(self, policy, p, q, step=0, **kw):
        mean, S = p
        old_mean, S_old = q
        d = lambda t: t.diagonal(dim1=-2, dim2=-1) if t.dim() == 3 else t
        B, A = mean.shape
        dev = mean.device
        batch = {"action": mean.detach(), "loc": old_mean, "var": d(S_old), "sample_log_prob": torch.zeros(B, device=dev),
                 "advantage": torch.zeros(B, device=dev)}
        out = ops.trpl_fwd_bwd(mean.detach().float(), d(S).detach().float().sqrt(), batch, None, mean_bound=self.mean_bound,
                               cov_bound=self.cov_bound, trust_region_coeff=self.trust_region_coeff, entropy_coef=0.0, critic_coef=0.0,
                               clip_value=0.0, global_batch=B, adv_stats=None, want_projection=True, proj_type=self.proj_code)
        pm, pv = out[5], out[6]
        return pm, (pv.diag_embed() if S.dim() == 3 else pv)


class FrobeniusProjectionLayer(KLProjectionLayer):
    """frob_projection_layer.py:9-88 on the diagonal policy (closed form; its regression loss is NOT detached from the projection)."""

    def __init__(self, proj_type="frob", **kw):
        super().__init__(proj_type="frob", **kw)


class WassersteinProjectionLayer(KLProjectionLayer):
    """w2_projection_layer.py:14-76 (commutative W2, precision-scaled) on the diagonal policy."""

    def __init__(self, proj_type="w2", **kw):
        super().__init__(proj_type="w2", **kw)


class _InjectGrad(torch.autograd.Function):
    """Scalar loss whose value and input gradients were already produced by the fused kernel: forward returns the value,
    backward hands out g * (precomputed gradients).  Actor and critic get separate nodes so that
    ``actor_loss.backward(); critic_loss.backward()`` (train.py:304-305) works without retain_graph."""

    @staticmethod
    def forward(ctx, value, n_in, *tensors):
        ctx.save_for_backward(*tensors[n_in:])
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        grads = ctx.saved_tensors
        return (None, None) + tuple(d * g for d in grads) + (None,) * len(grads)


def adv_stats_local(m, batch, out):
    """Rank-local sums (sum, sum of squares; fp64) of the advantage for its batch normalisation (trpl.py:248-252)."""
    adv = batch["advantage"].reshape(-1).float().contiguous()
    hip.call("grl_adv_stats", adv, out, adv.numel())


def trpl_launch(m, loc, sigma, value, batch, adv_stats, sums=None, maxes=None):
    """One launch of the fused kernel on detached inputs: rank-local (sums, maxes) and the gradients of the (1/B_global-scaled)
    losses with respect to loc, sigma and value."""
    p = m.projection
    B = loc.shape[0]
    sums, maxes, dloc, dsigma, dvalue, _, _ = ops.trpl_fwd_bwd(
        loc.detach(), sigma.detach(), batch, value.detach() if value is not None else None, mean_bound=p.mean_bound,
        cov_bound=p.cov_bound, trust_region_coeff=p.trust_region_coeff,
        entropy_coef=m.entropy_coef if m.entropy_bonus else 0.0, critic_coef=m.critic_coef,
        clip_value=float(m.clip_value) if m.clip_value is not None else 0.0, global_batch=B * m.world_size, adv_stats=adv_stats,
        sums=sums, maxes=maxes, proj_type=getattr(p, "proj_code", 0))
    return sums, maxes, dloc, dsigma, dvalue


def loss_values(m, sums, maxes):
    """(actor loss, critic loss, metrics dict) from the globally reduced sums / maxes (trpl.py:280-321): one launch, the entries
    are views of its 14-float output."""
    ent_coef = m.entropy_coef if m.entropy_bonus else 0.0
    o = torch.empty(14, device=sums.device, dtype=torch.float32)
    hip.call("grl_trpl_loss_values", sums, maxes, float(ent_coef), o)
    metrics = {"loss_trust_region": o[2], "loss_entropy": o[3], "ESS": o[4], "kl": o[5], "constraint": o[13], "mean_constraint": o[6],
               "mean_constraint_max": o[7], "cov_constraint": o[8], "cov_constraint_max": o[9], "entropy": o[10],
               "entropy_diff": o[11], "loss_objective_value": o[12]}
    return o[0], o[1], metrics


def _run_trpl(m, loc, sigma, value, batch):
    """Fused kernel (+ the advantage statistics launch) with the data-parallel reductions; everything detached."""
    with torch.no_grad():
        B = loc.shape[0]
        stats = None
        if m.normalize_advantage and B * m.world_size > 1:
            stats = torch.zeros(2, device=loc.device, dtype=torch.float64)
            adv_stats_local(m, batch, stats)
            if m.group is not None:
                import torch.distributed as dist
                dist.all_reduce(stats, group=m.group)
        sums, maxes, dloc, dsigma, dvalue = trpl_launch(m, loc, sigma, value, batch, stats)
        if m.group is not None:
            import torch.distributed as dist
            dist.all_reduce(sums, group=m.group)
            dist.all_reduce(maxes, op=dist.ReduceOp.MAX, group=m.group)
        actor, critic, metrics = loss_values(m, sums, maxes)
    actor = _InjectGrad.apply(actor, 2, loc, sigma, dloc, dsigma)
    if value is not None:
        critic = _InjectGrad.apply(critic, 1, value, dvalue.reshape(value.shape))
    return actor, critic, metrics


class TRPLLoss(nn.Module):
    """trpl.py:105-321.  ``actor_network`` is a GNNGaussianPolicyDiag (the reference digs the same module out of the
    ProbabilisticActor, trpl.py:243); ``critic_network`` a GNNVFNet/BaseCritic.  ``forward(batch)`` takes a mapping with the
    reference tensordict keys (observation groups, action, loc, covariance_matrix or var, sample_log_prob, advantage,
    value_target, state_value) and returns the loss dict consumed by examples/torchrl/train.py:280-301:
    loss_objective + loss_entropy + loss_trust_region carries the actor gradient, loss_critic the critic gradient."""

    def __init__(self, actor_network, critic_network, *, projection: KLProjectionLayer, clip_epsilon=0.2, entropy_bonus=True,
                 samples_mc_entropy=1, entropy_coef=0.01, critic_coef=1.0, trust_region_coef=1.0, loss_critic_type="l2",
                 normalize_advantage=True, gamma=None, separate_losses=False, clip_value=None, in_features=None, group=None, **kwargs):
        super().__init__()
        if loss_critic_type != "l2":
            raise NotImplementedError("loss_critic_type is l2 in configs/algorithm/objective/trpl.yaml:12")
        self.actor_network, self.critic_network, self.projection = actor_network, critic_network, projection
        self.entropy_bonus, self.entropy_coef, self.critic_coef = entropy_bonus, float(entropy_coef), float(critic_coef)
        self.normalize_advantage, self.clip_value = normalize_advantage, clip_value
        self.in_features = in_features or actor_network.hyper_data.spec.in_features
        self.group = group
        self._global_steps = 0

    @property
    def world_size(self):
        if self.group is None:
            return 1
        import torch.distributed as dist
        return dist.get_world_size(self.group)

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        b = dict(batch)
        if "var" not in b:
            b["var"] = b["covariance_matrix"].diagonal(dim1=-2, dim2=-1).contiguous()
        obs = [b[k] for k in self.in_features]
        loc, sigma = self.actor_network.forward_diag(*obs, train=True)
        value = self.critic_network(*obs) if self.critic_coef else None
        actor, critic, mt = _run_trpl(self, loc, sigma, value, b)
        out = {
            "loss_objective": actor - (mt["loss_trust_region"] + mt["loss_entropy"]),  # value = objective; gradient = d(actor loss)
            "loss_critic": critic, "loc": loc, "sigma": sigma, "state_value": value,
        }
        out.update({k: v for k, v in mt.items() if k != "loss_objective_value"})
        return out
