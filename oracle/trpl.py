"""Oracle: diag-Gaussian policy head, KL trust-region projection, TRPL loss, GAE (plain torch, CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Abbreviations (all under geometry_rl/algorithms/trust_region_projections/):
  diag.py   = models/policy/gnn_gaussian_policy_diag.py
  absg.py   = models/policy/abstract_gaussian_policy.py
  base.py   = projections/base_projection_layer.py
  kl.py     = projections/kl_projection_layer.py
  putils.py = utils/projection_utils.py
  trpl.py   = objectives/trpl.py
  outils.py = objectives/utils.py

NB the reference hands the policy's ``covariance_matrix`` (= diag(sigma^2)) to projection code that
treats it as a *std* matrix (trpl.py:241-245 -> kl.py:62-72).  This file reproduces that: every
function below that takes ``S`` receives the DIAGONAL of what the reference calls "std" there,
i.e. sigma^2 of the policy.

PARITY UNPINNED pieces: ``project_cov_diag_kl`` restates ITPAL's BatchedDiagCovOnlyProjection
(un-vendored C++, README.md:21-26) from its KKT conditions; ``gae_shifted`` restates torchrl 0.3.1
GAE(shifted=True) from the call site examples/torchrl/train.py:134-140,249-251.
"""
import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------- policy head
def inverse_softplus(x: torch.Tensor) -> torch.Tensor:
    """utils/torch_utils.py:361-370."""
    return (x.exp() - 1.0).log()


def std_head(hidden, w, b, init_std=1.0, minimal_std=1e-5, batch_size=None):
    """diag.py:65-87 + absg.py:124-134: sigma = softplus(hidden W^T + b + shift) + min_std, reshaped [B, A]."""
    shift = inverse_softplus(torch.tensor(init_std - minimal_std, dtype=hidden.dtype))
    std = F.softplus(F.linear(hidden, w, b) + shift) + minimal_std
    return std.reshape(batch_size, -1)


# --------------------------------------------------------------------------- diag-Gaussian helpers ("std"-matrix API)
def maha(mean, mean_other, S_other):
    """diag.py:128-131."""
    return ((mean - mean_other) / S_other).pow(2).sum(-1)


def log_determinant(S):
    """diag.py:117-126."""
    return 2 * S.log().sum(-1)


def entropy_std(S):
    """diag.py:111-115."""
    k = S.shape[-1]
    return 0.5 * (k * math.log(2 * math.e * math.pi) + log_determinant(S))


def gaussian_kl(p, q):
    """putils.py:34-67 with a diagonal policy: (mean part, cov part)."""
    mean, S = p
    mean_o, S_o = q
    k = mean.shape[-1]
    maha_part = 0.5 * maha(mean, mean_o, S_o)
    trace_part = (S / S_o).pow(2).sum(-1)  # trace_square(solve_triangular(S_o, S)) for diagonal matrices
    cov_part = 0.5 * (trace_part - k + log_determinant(S_o) - log_determinant(S))
    return maha_part, cov_part


def mean_projection(mean, old_mean, maha_part, eps):
    """base.py:71-100."""
    mask = maha_part > eps
    if mask.any():
        omega = torch.ones_like(maha_part)
        omega[mask] = torch.sqrt(maha_part[mask] / eps) - 1.0
        omega = torch.max(-omega, omega)[..., None]
        m = (mean + omega * old_mean) / (1 + omega + 1e-16)
        return torch.where(mask[..., None], m, mean)
    return mean


# --------------------------------------------------------------------------- ITPAL diag-cov KL projection (restated)
def _kl_of_eta(eta, t, o):
    """KL_cov(v(eta) || o) with 1/v = (eta/o + 1/t)/(eta+1); all float64, eta [B,1]."""
    v = (eta + 1.0) / (eta / o + 1.0 / t)
    return 0.5 * (v / o - 1.0 - v.log() + o.log()).sum(-1), v


def solve_eta(t: torch.Tensor, o: torch.Tensor, eps: float, iters: int = 100) -> torch.Tensor:
    """Per-sample eta >= 0 with KL_cov(eta) = eps (0 where the target already satisfies the bound).
    KL(eta) is monotone decreasing; bracket by doubling then bisect in float64."""
    t64, o64 = t.double(), o.double()
    kl0, _ = _kl_of_eta(torch.zeros(t.shape[0], 1, dtype=torch.float64), t64, o64)
    active = kl0 > eps
    lo = torch.zeros(t.shape[0], 1, dtype=torch.float64)
    hi = torch.ones(t.shape[0], 1, dtype=torch.float64)
    for _ in range(200):
        klh, _ = _kl_of_eta(hi, t64, o64)
        grow = (klh > eps) & active
        if not grow.any():
            break
        lo = torch.where(grow[:, None], hi, lo)
        hi = torch.where(grow[:, None], hi * 2.0, hi)
    for _ in range(iters):
        mid = 0.5 * (lo + hi)
        klm, _ = _kl_of_eta(mid, t64, o64)
        above = (klm > eps)[:, None]
        lo = torch.where(above, mid, lo)
        hi = torch.where(above, hi, mid)
    eta = 0.5 * (lo + hi)
    return torch.where(active[:, None], eta, torch.zeros_like(eta))


class ProjectCovDiagKL(torch.autograd.Function):
    """kl.py:162-204 KLProjectionGradFunctionDiagCovOnly around ITPAL BatchedDiagCovOnlyProjection
    [upstream, un-vendored]: forward solves the 1-D dual, backward differentiates through the KKT system
    (SURVEY Appendix A).  Inputs/outputs are diagonals [B, A]."""

    @staticmethod
    def forward(ctx, t, o, eps):
        eta = solve_eta(t.detach(), o.detach(), float(eps))
        _, v = _kl_of_eta(eta, t.detach().double(), o.detach().double())
        ctx.save_for_backward(t.detach().double(), o.detach().double(), eta, v)
        return v.to(t.dtype)

    @staticmethod
    def backward(ctx, dv):
        t, o, eta, v = ctx.saved_tensors
        dv64 = dv.double()
        dv_dt = v * v / (t * t * (eta + 1.0))
        dv_deta = -v * v * (1.0 / o - 1.0 / t) / (eta + 1.0) ** 2
        g = 0.5 * (1.0 / o - 1.0 / v)
        denom = (g * dv_deta).sum(-1, keepdim=True)
        active = eta > 0
        safe = torch.where(active, denom, torch.ones_like(denom))
        deta_dt = -g * dv_dt / safe
        corr = (dv64 * dv_deta).sum(-1, keepdim=True) * deta_dt
        dt = torch.where(active, dv64 * dv_dt + corr, dv64)
        return dt.to(dv.dtype), None, None


def project_cov_diag_kl(S, S_old, eps_cov):
    """kl.py:60-72: cov = S**2, old_cov = S_old**2, proj = sqrt(ITPAL(cov, old_cov, eps))."""
    proj_cov = ProjectCovDiagKL.apply(S.pow(2), S_old.pow(2), eps_cov)
    return proj_cov.sqrt()


def kl_projection(p, q, mean_bound, cov_bound):
    """kl.py:15-111 (contextual, diagonal) followed by the identity entropy projection
    (base.py:232-273 with bound = -inf, putils.py:280)."""
    mean, S = p
    old_mean, S_old = q
    mean_part, _ = gaussian_kl(p, q)
    proj_mean = mean_projection(mean, old_mean, mean_part, mean_bound)
    proj_S = project_cov_diag_kl(S, S_old, cov_bound)
    return proj_mean, proj_S


# --------------------------------------------------------------------------- Frobenius / Wasserstein projections (diagonal policy)
def frobenius_value(p, q):
    """putils.py:70-104 (scale_prec=True): (maha(mean, mean_o, S_o), |S_o^2 - S^2|_F^2)."""
    (mean, S), (mean_o, S_o) = p, q
    return maha(mean, mean_o, S_o), (S_o.pow(2) - S.pow(2)).pow(2).sum(-1)


def wasserstein_value(p, q):
    """putils.py:107-149 (commutative, scale_prec=True): (maha, tr(I + S_o^-1 S^2 S_o^-1 - 2 S_o^-1 S)) = (maha, sum (1 - S/S_o)^2)."""
    (mean, S), (mean_o, S_o) = p, q
    return maha(mean, mean_o, S_o), (1.0 - S / S_o).pow(2).sum(-1)


def _eta_from_part(cov_part, eps_cov):
    """frob_projection_layer.py:49-56 / w2_projection_layer.py:58-64: eta = |sqrt(part/eps) - 1| where the bound is violated, 1 elsewhere
    (those rows are masked out afterwards)."""
    mask = cov_part > eps_cov
    eta = torch.ones_like(cov_part)
    eta[mask] = torch.sqrt(cov_part[mask] / eps_cov) - 1.0
    return mask, torch.max(-eta, eta)


def frobenius_projection(p, q, mean_bound, cov_bound):
    """frob_projection_layer.py:10-63 (+ identity entropy projection): new_cov = (S^2 + eta S_o^2) / (1 + eta), proj_S = chol = sqrt."""
    (mean, S), (mean_o, S_o) = p, q
    mean_part, cov_part = frobenius_value(p, q)
    proj_mean = mean_projection(mean, mean_o, mean_part, mean_bound)
    mask, eta = _eta_from_part(cov_part, cov_bound)
    if mask.any():
        new_cov = (S.pow(2) + eta[..., None] * S_o.pow(2)) / (1.0 + eta + 1e-16)[..., None]
        proj_S = torch.where(mask[..., None], new_cov.sqrt(), S)
    else:
        proj_S = S
    return proj_mean, proj_S


def wasserstein_projection(p, q, mean_bound, cov_bound):
    """w2_projection_layer.py:15-68: new_sqrt = (S + eta S_o) / (1 + eta)."""
    (mean, S), (mean_o, S_o) = p, q
    mean_part, cov_part = wasserstein_value(p, q)
    proj_mean = mean_projection(mean, mean_o, mean_part, mean_bound)
    mask, eta = _eta_from_part(cov_part, cov_bound)
    if mask.any():
        new_S = (S + eta[..., None] * S_o) / (1.0 + eta + 1e-16)[..., None]
        proj_S = torch.where(mask[..., None], new_S, S)
    else:
        proj_S = S
    return proj_mean, proj_S


def frobenius_trust_region_loss(p, proj_p, coeff):
    """frob_projection_layer.py:73-88 (contextual std): maha(mean, proj_mean, S) + |S - proj_S|^2, proj_p NOT detached."""
    mean_diff = maha(p[0], proj_p[0], p[1])
    cov_diff = (p[1] - proj_p[1]).pow(2).sum(-1)
    return (mean_diff + cov_diff).mean() * coeff


def wasserstein_trust_region_loss(p, proj_p, coeff):
    """base.py:292-327 with trust_region_value = w2 value: (p, stopgrad(proj_p))."""
    m_d, c_d = wasserstein_value(p, (proj_p[0].detach(), proj_p[1].detach()))
    return (m_d + c_d).mean() * coeff


PROJECTIONS = {"kl": (kl_projection, gaussian_kl), "frob": (frobenius_projection, frobenius_value), "w2": (wasserstein_projection, wasserstein_value)}


# --------------------------------------------------------------------------- TRPL loss
def mvn_diag_log_prob(x, mean, var):
    """torch.distributions.MultivariateNormal(mean, diag(var)).log_prob (trpl.py:245-246)."""
    k = x.shape[-1]
    return -0.5 * (((x - mean) ** 2 / var).sum(-1) + k * LOG_2PI + var.log().sum(-1))


def mvn_diag_entropy(var):
    """MultivariateNormal.entropy for a diagonal covariance (trpl.py:309-312)."""
    k = var.shape[-1]
    return 0.5 * (k * (1.0 + LOG_2PI) + var.log().sum(-1))


def clipped_value_loss(value, old_value, target, clip_value):
    """trpl.py:213-228 + outils.py:5-28 (loss_critic_type = l2)."""
    loss = (value - target) ** 2
    if clip_value:
        clipped = old_value + (value - old_value).clamp(-clip_value, clip_value)
        loss = torch.max(loss, (clipped - target) ** 2)
    return loss


def trpl_loss(
    loc,
    var,
    batch: Dict[str, torch.Tensor],
    state_value,
    *,
    mean_bound,
    cov_bound,
    trust_region_coeff,
    entropy_coef,
    critic_coef,
    clip_value=0.2,
    normalize_advantage=True,
    adv_stats=None,
    proj_type="kl",
) -> Dict[str, torch.Tensor]:
    """trpl.py:275-321 TRPLLoss.forward.

    loc [B,A], var [B,A] (= diagonal of the policy's ``covariance_matrix``) come from the current policy;
    ``batch`` holds action, loc (old), var (old diag covariance), sample_log_prob, advantage, value_target,
    state_value (old).  ``adv_stats`` = (mean, unbiased std) overrides the in-batch statistics (used to check the
    data-parallel shards against the global batch).
    """
    adv = batch["advantage"]
    if normalize_advantage and adv.numel() > 1:  # trpl.py:286-289
        if adv_stats is None:
            a_loc, a_scale = adv.mean(), adv.std().clamp_min(1e-6)
        else:
            a_loc, a_scale = adv_stats
        adv = (adv - a_loc) / a_scale
    p = (loc, var)  # trpl.py:241 (covariance diagonal used as "std")
    q = (batch["loc"], batch["var"])
    project, tr_value = PROJECTIONS[proj_type]
    proj_mean, proj_S = project(p, q, mean_bound, cov_bound)  # trpl.py:244
    log_prob = mvn_diag_log_prob(batch["action"], proj_mean, proj_S)  # trpl.py:245-246 (proj_S as covariance)
    lw = log_prob - batch["sample_log_prob"]
    with torch.no_grad():  # trpl.py:294-300
        ess = (2 * lw.logsumexp(0) - (2 * lw).logsumexp(0)).exp() / lw.shape[0]
    out = {"loss_objective": -(lw.exp() * adv.reshape(-1)).mean()}  # trpl.py:302-303
    # base.py:292-327: gaussian_kl(p, stopgrad(proj_p)), summed parts, mean, times coefficient
    if proj_type == "frob":
        out["loss_trust_region"] = frobenius_trust_region_loss(p, (proj_mean, proj_S), trust_region_coeff)
    else:
        m_d, c_d = tr_value(p, (proj_mean.detach(), proj_S.detach()))
        out["loss_trust_region"] = (m_d + c_d).mean() * trust_region_coeff
    ent = mvn_diag_entropy(proj_S)  # trpl.py:309-312
    out["entropy_dist"] = ent.mean().detach()
    out["loss_entropy"] = -entropy_coef * ent.mean()
    out["loss_critic"] = (
        critic_coef
        * clipped_value_loss(state_value.reshape(-1), batch["state_value"].reshape(-1),
                             batch["value_target"].reshape(-1), clip_value)
    ).mean()
    out["ESS"] = ess
    with torch.no_grad():  # trpl.py:255-273 -> base.py:332-384 called with (p, proj_p)
        pq = (proj_mean, proj_S)
        km, kc = gaussian_kl(p, pq)
        mk, ck = tr_value(p, pq)
        e_old, e_new = entropy_std(proj_S), entropy_std(var)
        out.update(
            kl=(km + kc).mean(), constraint=(mk + ck).mean(), mean_constraint=mk.mean(), mean_constraint_max=mk.max(),
            cov_constraint=ck.mean(), cov_constraint_max=ck.max(), entropy=e_new.mean(),
            entropy_diff=(e_old - e_new).mean(),
        )
    out["proj_mean"], out["proj_S"] = proj_mean, proj_S
    return out


# --------------------------------------------------------------------------- GAE
def gae_shifted(reward, done, terminated, values, gamma=0.99, lmbda=0.95):
    """torchrl 0.3.1 GAE(shifted=True, average_gae=False) [upstream] as called at train.py:134-140,249-251.

    reward/done/terminated [N,T], values [N,T+1] (critic on [obs_0..obs_{T-1}, next_obs_{T-1}]).
    delta_t = r_t + gamma (1-term_t) V_{t+1} - V_t ; A_t = delta_t + gamma lambda (1-done_t) A_{t+1}.
    Returns (advantage, value_target) [N,T]."""
    v, nv = values[:, :-1], values[:, 1:]
    not_done = 1.0 - done.to(reward.dtype)
    not_term = 1.0 - terminated.to(reward.dtype)
    delta = reward + gamma * not_term * nv - v
    adv = torch.zeros_like(reward)
    run = torch.zeros_like(reward[:, 0])
    for t in range(reward.shape[1] - 1, -1, -1):
        run = delta[:, t] + gamma * lmbda * not_done[:, t] * run
        adv[:, t] = run
    return adv, adv + v
