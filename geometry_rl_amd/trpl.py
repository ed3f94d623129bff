"""TRPL objective on the fused HIP kernel -- drop-in for
``geometry_rl/algorithms/trust_region_projections/objectives/trpl.py`` (TRPLLoss) and
``.../projections/kl_projection_layer.py`` (KLProjectionLayer, diagonal / contextual case).

The reference moves (mean, covariance) to the CPU, calls ITPAL per sample and comes back (trpl.py:241-245).  Here the
projection, log-ratio objective, trust-region regression, entropy bonus, clipped value loss, all metrics AND their
analytic gradients come out of one launch of ``grl_trpl_fwd_bwd``; nothing leaves the device and there is no host sync."""
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import hip, ops

try:   # torchrl / tensordict are optional: with them the loss module IS a torchrl LossModule and returns a TensorDict
    from tensordict import TensorDict as _TensorDict
except Exception:   # pragma: no cover - not installed in the build image
    _TensorDict = None
try:
    from torchrl.objectives import LossModule as _LossBase
except Exception:   # pragma: no cover
    _LossBase = nn.Module


class LossDict(dict):
    """What ``TRPLLoss.forward`` returns when ``tensordict`` is not installed: a dict of scalar tensors with the slice of the
    TensorDict protocol that examples/torchrl/train.py:279-316 uses on the loss output -- ``loss[key]``, ``loss.select(*keys)``,
    ``.detach()``, ``.get(key[, default])``, ``.set(key, value)``, ``.apply(fn)``, ``.items()`` / ``.keys()`` -- so that loop
    runs unchanged on either return type."""

    batch_size = torch.Size([])

    def select(self, *keys, strict: bool = True):
        missing = [k for k in keys if k not in self]
        if missing and strict:
            raise KeyError(f"keys {missing} not found in the loss output (has {sorted(self)})")
        return LossDict((k, self[k]) for k in keys if k in self)

    def exclude(self, *keys):
        return LossDict((k, v) for k, v in self.items() if k not in keys)

    def detach(self):
        return LossDict((k, v.detach() if torch.is_tensor(v) else v) for k, v in self.items())

    def clone(self, recurse: bool = True):
        return LossDict((k, v.clone() if (recurse and torch.is_tensor(v)) else v) for k, v in self.items())

    def set(self, key, value):
        self[key] = value
        return self

    def apply(self, fn, batch_size=None, **kw):
        return LossDict((k, fn(v)) for k, v in self.items())

    def to(self, *a, **k):
        return LossDict((key, v.to(*a, **k) if torch.is_tensor(v) else v) for key, v in self.items())

    def to_dict(self):
        return dict(self)


def _diag(t):
    return t.diagonal(dim1=-2, dim2=-1) if t.dim() == 3 else t


# ---- entropy control of the projection (base_projection_layer.py:14-68, projection_utils.py:252-280).  Plain tensor arithmetic on the
#      [B, A, A] "std" matrices of the reference's API (what the policy returns as covariance), differentiable: a host-side step on the
#      tensors that feed / leave the fused kernel.  Pinned by tests/golden/tier2e_std_entropy.npz (reference code).
def _entropy_of(p):
    """gnn_gaussian_policy_diag.py:104-126 on p = (mean, std matrix or its diagonal)."""
    import math
    d = _diag(p[1])
    return 0.5 * (d.shape[-1] * math.log(2 * math.e * math.pi) + 2 * d.log().sum(-1))


def entropy_inequality_projection(policy, p, beta):
    """base_projection_layer.py:14-44: samples whose entropy is below ``beta`` get their std scaled by exp((beta - entropy) / k)."""
    mean, std = p
    k = std.shape[-1]
    ent = policy.entropy(p) if policy is not None else _entropy_of(p)
    mask = ent < beta
    if not bool(mask.any()):
        return p
    beta = torch.as_tensor(beta, dtype=ent.dtype, device=ent.device).expand_as(ent)
    alpha = torch.where(mask, ((beta - ent) / k).exp(), torch.ones_like(ent))
    return mean, std * alpha.reshape(alpha.shape + (1,) * (std.dim() - alpha.dim()))


def entropy_equality_projection(policy, p, beta):
    """base_projection_layer.py:47-68: every sample's std is scaled so that its entropy EQUALS ``beta``."""
    mean, std = p
    k = std.shape[-1]
    ent = policy.entropy(p) if policy is not None else _entropy_of(p)
    alpha = ((beta - ent) / k).exp()
    return mean, std * alpha.reshape(alpha.shape + (1,) * (std.dim() - alpha.dim()))


def get_entropy_schedule(schedule_type, total_train_steps, dim):
    """projection_utils.py:252-280: f(initial_entropy, target_entropy, temperature, step) -> entropy bound."""
    if schedule_type == "linear":
        return lambda initial, target, temperature, step: step * (target - initial) / total_train_steps + initial
    if schedule_type == "exp":
        return lambda initial, target, temperature, step: dim * target + (initial - dim * target) * temperature ** (10 * step / total_train_steps)
    return lambda initial, target, temperature, step: torch.as_tensor(float("-inf"))


class KLProjectionLayer:
    """Hyper-parameter holder with the reference constructor (base_projection_layer.py:123-197; configs/algorithm/projection/kl.yaml).
    ``__call__(policy, p, q, step)`` returns the projected (mean, "std") like the reference layer, for inspection."""

    def __init__(self, proj_type="kl", mean_bound=0.0, cov_bound=0.0, trust_region_coeff=0.0, scale_prec=True, mean_eq=False,
                 entropy_schedule=None, action_dim=None, total_train_steps=None, target_entropy=0.0, temperature=0.0,
                 entropy_eq=False, entropy_first=False, cpu=False, dtype=torch.float32, **ignored):
        # entropy control (base_projection_layer.py:176-185): the schedule, the bound and the two projections are here (pinned by the
        # tier-2e fixture); the FUSED update kernel does not apply them -- TRPLLoss refuses a layer with an active schedule instead of
        # silently skipping the entropy projection (no reference TRPL config turns it on: configs/algorithm/projection/kl.yaml:9)
        if entropy_schedule and not (action_dim and total_train_steps):
            raise AssertionError("entropy_schedule needs action_dim and total_train_steps (base_projection_layer.py:177)")
        self.entropy_schedule_type = entropy_schedule or None
        self._entropy_schedule = get_entropy_schedule(self.entropy_schedule_type, total_train_steps, dim=action_dim)
        self._entropy_proj = entropy_equality_projection if entropy_eq else entropy_inequality_projection
        self.target_entropy, self.temperature = float(target_entropy), float(temperature)
        self.entropy_first, self.entropy_eq = bool(entropy_first), bool(entropy_eq)
        kinds = {"kl": 0, "frob": 1, "frobenius": 1, "w2": 2, "wasserstein": 2}
        if proj_type.lower() not in kinds or mean_eq or not scale_prec:
            raise NotImplementedError("projections: kl | frob | w2 (commutative), each with the Mahalanobis (scale_prec) mean bound")
        self.proj_type, self.mean_bound, self.cov_bound = proj_type, float(mean_bound), float(cov_bound)
        self.proj_code = kinds[proj_type.lower()]
        self.trust_region_coeff = float(trust_region_coeff)
        self.initial_entropy = None

    def get_entropy_bound(self, step):
        """base_projection_layer.py:329-330."""
        init = self.initial_entropy if self.initial_entropy is not None else torch.as_tensor(0.0)
        return self._entropy_schedule(init, torch.as_tensor(self.target_entropy, dtype=init.dtype, device=init.device), self.temperature, step)

    def entropy_projection(self, policy, p, q, step):
        """The entropy half of base_projection_layer.py:200-205,266-283 on its own: latches the initial entropy (mean entropy of the OLD
        distribution at the first call) and projects ``p`` onto the scheduled bound."""
        if self.initial_entropy is None:
            self.initial_entropy = (policy.entropy(q) if policy is not None else _entropy_of(q)).mean().detach()
        return self._entropy_proj(policy, p, self.get_entropy_bound(step) * p[0].new_ones(p[0].shape[0]))

    def __call__(self, policy, p, q, step=0, **kw):
        if self.entropy_schedule_type and self.entropy_first:     # base_projection_layer.py:266-283: entropy first, then the trust region
            p = self.entropy_projection(policy, p, q, step)
        out = self._trust_region_call(policy, p, q)
        if self.entropy_schedule_type and not self.entropy_first:
            out = self.entropy_projection(policy, out, q, step)
        return out

    def _trust_region_call(self, policy, p, q):
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

    # ---- base_projection_layer.py:292-327.  p = (mean, S) live (S = what the policy returns as covariance: the layer's "std"),
    #      proj_p = the projection's output, treated as a constant.  Value AND gradient come from the fused kernel with its projection
    #      step skipped (grl_trpl_target_terms); autograd sees one node whose backward hands the kernel's gradients to p.
    def _target_terms(self, p, target):
        mean, S = p
        Sd, tm, tS = _diag(S), target[0].detach(), _diag(target[1]).detach()
        B = mean.shape[0]
        sigma = Sd.detach().float().sqrt()
        sums, maxes, dloc, dsigma = ops.trpl_target_terms(mean.detach().float(), sigma, tm.float().contiguous(), tS.float().contiguous(),
                                                          mean_bound=self.mean_bound, cov_bound=self.cov_bound,
                                                          trust_region_coeff=self.trust_region_coeff, global_batch=B,
                                                          proj_type=self.proj_code)
        return mean, Sd, sigma, sums, maxes, dloc, dsigma

    def get_trust_region_loss(self, policy, p, proj_p):
        mean, Sd, sigma, sums, maxes, dloc, dsigma = self._target_terms(p, proj_p)
        value = (sums[1] / sums[10]).float()
        if not (mean.requires_grad or Sd.requires_grad):
            return value
        dS = dsigma / (2.0 * sigma)   # the kernel differentiates with respect to sigma = sqrt(S)
        return _InjectGrad.apply(value, 2, mean, Sd, dloc, dS)

    def compute_metrics(self, policy, p, q, step=None, aggregate=True):
        """base_projection_layer.py:332-384 (aggregate=True): means and maxima of the projection's own measure of (p, q), the KL, and the
        entropies.  The *_max entries the loss module never reads (kl_max, constraint_max, entropy_max, entropy_diff_max) are not produced."""
        if not aggregate:
            raise NotImplementedError("per-sample metrics (aggregate=False) are not used on the policy-update path")
        with torch.no_grad():
            _, _, _, sums, maxes, _, _ = self._target_terms((p[0].detach(), p[1].detach()), q)
            n = sums[10]
            mx = maxes.view(torch.float32)
            mc, cc = (sums[6] / n).float(), (sums[7] / n).float()
            return OrderedDict(kl=(sums[11] / n).float(), constraint=mc + cc, mean_constraint=mc, cov_constraint=cc,
                               entropy=(sums[8] / n).float(), entropy_diff=(sums[9] / n).float(), mean_constraint_max=mx[0],
                               cov_constraint_max=mx[1])


class FrobeniusProjectionLayer(KLProjectionLayer):
    """frob_projection_layer.py:9-88 on the diagonal policy (closed form; its regression loss is NOT detached from the projection)."""

    def __init__(self, proj_type="frob", **kw):
        super().__init__(proj_type="frob", **kw)


class WassersteinProjectionLayer(KLProjectionLayer):
    """w2_projection_layer.py:14-76 (commutative W2, precision-scaled) on the diagonal policy."""

    def __init__(self, proj_type="w2", **kw):
        super().__init__(proj_type="w2", **kw)


def get_projection_layer(proj_type: str = "", **kwargs) -> KLProjectionLayer:
    """projection_factory.py:9-48 -- the call builders/utils_algo_graph.py:246-253 makes (``action_dim``, ``total_train_steps``, ``cpu``,
    ``dtype`` + the entries of configs/algorithm/projection/<type>.yaml as keyword arguments).  "kl" / "frob" / "w2": the fused-kernel
    layers of this module; the reference's other branches are outside the TRPL hot path (PPO objective, PAPI,
    non-commuting W2) and say so; anything else is the reference's ValueError."""
    key = (proj_type or "").lower()
    if key == "kl":
        return KLProjectionLayer(proj_type, **kwargs)
    if key == "frob":
        return FrobeniusProjectionLayer(proj_type, **kwargs)
    if key == "w2":
        return WassersteinProjectionLayer(proj_type, **kwargs)
    if not key or key.isspace() or key in ("ppo", "sac", "td3", "mpo", "vlearn", "vtrace", "awr", "entropy", "w2_non_com", "papi"):
        raise NotImplementedError(f"projection '{proj_type}': only the TRPL projections kl | frob | w2 are built (the PPO-style "
                                  "BaseProjectionLayer, PAPI and the non-commuting W2 are outside the policy-update hot path)")
    raise ValueError(f"Invalid projection type {proj_type}. Choose one of None/' ', 'ppo', 'sac', 'td3', 'mpo', 'vtrace', 'papi', 'w2', "
                     "'w2_non_com', 'frob', 'kl', or 'entropy'.")


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


def trpl_launch(m, loc, sigma, value, batch, adv_stats, sums=None, maxes=None, defer_fold=False, adv_local=False):
    """One launch of the fused kernel on detached inputs: rank-local (sums, maxes) and the gradients of the (1/B_global-scaled)
    losses with respect to loc, sigma and value.  ``defer_fold``: see ops.trpl_fwd_bwd (``sums`` comes back as the folding callable).
    ``adv_local`` (one rank): the advantage statistics are summed inside the kernel; ``value=None``: actor-only (the critic's share of
    the loss comes from ``value_loss`` on the critic's lane)."""
    p = m.projection
    B = loc.shape[0]
    sums, maxes, dloc, dsigma, dvalue, _, _ = ops.trpl_fwd_bwd(
        loc.detach(), sigma.detach(), batch, value.detach() if value is not None else None, mean_bound=p.mean_bound,
        cov_bound=p.cov_bound, trust_region_coeff=p.trust_region_coeff,
        entropy_coef=m.entropy_coef if m.entropy_bonus else 0.0, critic_coef=m.critic_coef,
        clip_value=float(m.clip_value) if m.clip_value is not None else 0.0, global_batch=B * m.world_size, adv_stats=adv_stats,
        sums=sums, maxes=maxes, proj_type=getattr(p, "proj_code", 0), defer_fold=defer_fold, adv_local=adv_local)
    return sums, maxes, dloc, dsigma, dvalue


def value_loss(m, value, batch):
    """The critic's share of the loss on its own (clipped l2 value loss, trpl.py:213-228): -> (dvalue [B], loss_critic float32 0-d, sums
    fp64[2] = summed loss and its mean over the global batch).  One launch; one rank (nothing is all-reduced)."""
    import ctypes
    B = value.shape[0]
    dev = value.device
    dvalue = torch.empty(B, device=dev, dtype=torch.float32)
    out2 = torch.empty(2, device=dev, dtype=torch.float64)
    mean = torch.empty(1, device=dev, dtype=torch.float32)
    hip.call("grl_value_loss", value.reshape(B).contiguous(), batch["state_value"].reshape(B).contiguous(),
             batch["value_target"].reshape(B).contiguous(), ctypes.c_double(float(m.clip_value) if m.clip_value is not None else 0.0),
             ctypes.c_double(float(m.critic_coef)), ctypes.c_double(1.0 / (B * m.world_size)), dvalue, out2, mean, B)
    return dvalue, mean[0], out2


def report_dict(o):
    """The 14-float output of grl_trpl_report / grl_fold_adam_report as (actor loss, metrics dict of views)."""
    return o[0], {"loss_trust_region": o[2], "loss_entropy": o[3], "ESS": o[4], "kl": o[5], "constraint": o[13], "mean_constraint": o[6],
                  "mean_constraint_max": o[7], "cov_constraint": o[8], "cov_constraint_max": o[9], "entropy": o[10],
                  "entropy_diff": o[11], "loss_objective_value": o[12]}


def report_values(m, slots, B, sums, maxes):
    """Fold of the fused kernel's per-workgroup slots and the reported values in ONE launch (one rank): -> (actor loss, critic loss,
    metrics dict) like ``loss_values``."""
    ent_coef = m.entropy_coef if m.entropy_bonus else 0.0
    o = torch.empty(14, device=sums.device, dtype=torch.float32)
    hip.call("grl_trpl_report", slots, B, sums, maxes, float(ent_coef), o)
    metrics = {"loss_trust_region": o[2], "loss_entropy": o[3], "ESS": o[4], "kl": o[5], "constraint": o[13], "mean_constraint": o[6],
               "mean_constraint_max": o[7], "cov_constraint": o[8], "cov_constraint_max": o[9], "entropy": o[10],
               "entropy_diff": o[11], "loss_objective_value": o[12]}
    return o[0], o[1], metrics


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


_TD_KEYS = ("action", "loc", "covariance_matrix", "var", "sample_log_prob", "advantage", "value_target", "state_value")


def _as_batch(td, in_features) -> Dict[str, torch.Tensor]:
    """A plain dict of the tensors the loss reads, from a Mapping or any object with the TensorDict ``.get(key[, default])``."""
    if isinstance(td, dict):
        return dict(td)
    out = {}
    for k in tuple(dict.fromkeys(tuple(in_features))) + _TD_KEYS:
        try:
            v = td.get(k, None)
        except TypeError:
            v = td.get(k) if k in td.keys() else None
        if v is not None:
            out[k] = v
    return out


class TRPLLoss(_LossBase):
    """trpl.py:105-321.  ``actor_network`` is a GNNGaussianPolicyDiag (the reference digs the same module out of the
    ProbabilisticActor, trpl.py:243: ``actor_network.get_submodule("0").module`` is tried first, so a ProbabilisticActor wrapping
    the policy is accepted too); ``critic_network`` a GNNVFNet/BaseCritic (or a ValueOperator around one).
    ``forward(tensordict)`` takes a TensorDict -- or any mapping / object with ``.get(key)`` -- holding the reference keys
    (observation groups, action, loc, covariance_matrix or var, sample_log_prob, advantage, value_target, state_value) and returns
    a TensorDict when ``tensordict`` is installed, else a :class:`LossDict` with the same access protocol, so
    examples/torchrl/train.py:279-316 runs on it unchanged: ``loss.select(*loss_types).detach()``,
    ``loss["loss_objective"] (+= loss_entropy, loss_trust_region)`` carries the actor gradient, ``loss["loss_critic"]`` the critic
    gradient.  With torchrl importable the class is a ``torchrl.objectives.LossModule``."""

    def __init__(self, actor_network, critic_network, *, projection: KLProjectionLayer, clip_epsilon=0.2, entropy_bonus=True,
                 samples_mc_entropy=1, entropy_coef=0.01, critic_coef=1.0, trust_region_coef=1.0, loss_critic_type="l2",
                 normalize_advantage=True, gamma=None, separate_losses=False, clip_value=None, in_features=None, group=None,
                 critic_in_features=None, **kwargs):
        super().__init__()
        if loss_critic_type != "l2":
            raise NotImplementedError("loss_critic_type is l2 in configs/algorithm/objective/trpl.yaml:12")
        def unwrap(m, attr):   # ProbabilisticActor(TensorDictModule(policy)) / ValueOperator(critic): the wrapped nn.Module
            for path in ("0.module", "module.0.module", "module"):
                try:
                    inner = m.get_submodule(path)
                except Exception:
                    continue
                if hasattr(inner, attr):
                    return inner
            return m
        actor_network = unwrap(actor_network, "forward_diag")
        critic_network = unwrap(critic_network, "_network1")
        if getattr(projection, "entropy_schedule_type", None):
            raise NotImplementedError("an entropy schedule (base_projection_layer.py:266-283) is not applied by the fused update kernel; the "
                                      "projection layer offers it on its own (KLProjectionLayer.entropy_projection / __call__)")
        self.actor_network, self.critic_network, self.projection = actor_network, critic_network, projection
        self.trust_region_coef = trust_region_coef
        self.entropy_bonus, self.entropy_coef, self.critic_coef = entropy_bonus, float(entropy_coef), float(critic_coef)
        self.normalize_advantage, self.clip_value = normalize_advantage, clip_value
        # tensordict keys handed POSITIONALLY to the actor / the critic (the in_keys of their TensorDictModules,
        # utils_algo_graph.py:113-116,160-176).  They may differ: config 1 feeds its transformer actor the normalised vectors in the
        # raw-vector slots (configs/rigid_insertion_multi_transformer_trpl_cfg.yaml:88-94) while the critic reads the raw ones.
        self.in_features = list(in_features or actor_network.hyper_data.spec.in_features)
        self.critic_in_features = list(critic_in_features or self.in_features)
        self.group = group
        self._global_steps = 0

    @property
    def world_size(self):
        if self.group is None:
            return 1
        import torch.distributed as dist
        return dist.get_world_size(self.group)

    @property
    def out_keys(self):   # trpl.py:155-170 (+ the TRPL entries forward() sets, :302-321)
        keys = ["loss_objective", "loss_trust_region"]
        if self.entropy_bonus:
            keys += ["entropy", "loss_entropy"]
        if self.critic_coef:
            keys.append("loss_critic")
        return keys + ["ESS", "kl", "constraint", "mean_constraint", "mean_constraint_max", "cov_constraint", "cov_constraint_max",
                       "entropy_diff"]

    def forward(self, tensordict):
        b = _as_batch(tensordict, self.in_features + self.critic_in_features)
        if "var" not in b:
            b["var"] = b["covariance_matrix"].diagonal(dim1=-2, dim2=-1).contiguous()
        obs = [b[k] for k in self.in_features]
        loc, sigma = self.actor_network.forward_diag(*obs, train=True)
        value = self.critic_network(*[b[k] for k in self.critic_in_features]) if self.critic_coef else None
        actor, critic, mt = _run_trpl(self, loc, sigma, value, b)
        out = {
            "loss_objective": actor - (mt["loss_trust_region"] + mt["loss_entropy"]),  # value = objective; gradient = d(actor loss)
            "loss_critic": critic, "loc": loc, "sigma": sigma, "state_value": value,
        }
        out.update({k: v for k, v in mt.items() if k != "loss_objective_value"})
        if _TensorDict is not None:   # the reference's return type (trpl.py:302)
            extra = {k: out.pop(k) for k in ("loc", "sigma", "state_value")}
            td = _TensorDict(out, [])
            td.__dict__["_grl_outputs"] = extra
            return td
        return LossDict(out)
