"""Shared tolerance rules of the GPU parity tests.

Gradients: every gradient tensor is held to ``G_TOL`` = 2e-4 of ITS OWN largest reference entry -- not of max(1, .): a tensor whose largest
gradient is 1e-3 is checked at 2e-7, not at 2e-4.  The only floor is for tensors whose whole gradient is numerically nothing beside the
rest of the network (a bias whose gradient cancels to ~1e-9 where the weights next to it have 1e-2): their scale is at least
``NET_FLOOR`` = 1e-4 of the largest gradient entry of the network they belong to.  Measured (profiles/r04_parity_sizes.json): 4-7e-5 of the
tensor's own max |g| against an fp64 oracle, 2.5-9e-6 for the fp32 CPU oracle itself.

Parameters after Adam: with zero moments the first Adam step is  dp = -lr g / (|g| + eps),  d(dp)/dg = -lr eps / (|g| + eps)^2  -- an entry
whose gradient is small beside eps = 1e-5 turns an ABSOLUTE gradient error dg into dp = lr dg / eps = 30 dg (this, not the kernels, is what put
the cloth gate's `gnn.basis_fn.3.weight` at 1.45e-5 of a flat 2e-5 allowance in round 3: gradient entries near zero in a tensor whose
largest entry is ~1e-2, i.e. dg ~ 5e-7 = 5e-5 relative).  The bound is therefore derived from the gradient tolerance instead of being a
flat number.  Per tensor:  |dp| <= lr * min(2, G_TOL * scale / eps) + P_ROUND  (2 lr: a sign flip of a saturated entry; P_ROUND: fp32 rounding
of the parameter itself).  Per ENTRY (used wherever the reference gradient is at hand and nothing rescales it):
|dp_i| <= lr * min(2, dg eps / (max(|g_i| - dg, 0) + eps)^2) + P_ROUND with dg = G_TOL * scale -- entries with a large gradient are saturated
(dp ~ lr sign g) and held to fp32 rounding, only the entries near zero get the 30 x allowance."""
import torch

G_TOL = 2e-4
NET_FLOOR = 1e-4
P_ROUND = 4e-7   # x max(1, |p|): one to two units in the last place of the fp32 parameter itself (p - dp rounds on both sides)


def grad_scales(ref_grads):
    """{name: reference gradient} of ONE network -> {name: scale} with scale = max(max |g_t|, NET_FLOOR * max over the network)."""
    own = {k: (float(v.detach().abs().max()) if v.numel() else 0.0) for k, v in ref_grads.items()}
    net = max(own.values()) if own else 0.0
    return {k: max(v, NET_FLOOR * net) for k, v in own.items()}


def grad_error(got, ref):
    got, ref = torch.as_tensor(got).detach().cpu().double(), torch.as_tensor(ref).detach().cpu().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max()) if ref.numel() else 0.0


def adam_first_step_bound(lr, eps, grad_scale, clip=False, p_ref=None):
    """Largest parameter difference a gradient error of G_TOL * grad_scale can cause in the first Adam step (zero moments); with gradient
    clipping the clip coefficient carries the relative error of the gradient norm on top (twice the allowance).  ``p_ref``: the reference
    parameter (its magnitude scales the rounding allowance)."""
    dg = G_TOL * grad_scale * (2.0 if clip else 1.0)
    pmax = max(1.0, float(torch.as_tensor(p_ref).detach().abs().max())) if p_ref is not None and torch.as_tensor(p_ref).numel() else 1.0
    return lr * min(2.0, dg / eps) + P_ROUND * pmax


def adam_first_step_bound_elem(lr, eps, ref_grad, grad_scale, p_ref=None):
    """Entry-wise form of adam_first_step_bound: a tensor of allowances for |p_hip - p_ref| given the reference gradient of the tensor."""
    dg = G_TOL * grad_scale
    g = torch.as_tensor(ref_grad).detach().cpu().double().abs()
    sens = dg * eps / ((g - dg).clamp_min(0.0) + eps) ** 2
    rnd = P_ROUND * (torch.as_tensor(p_ref).detach().cpu().double().abs().clamp_min(1.0) if p_ref is not None else 1.0)
    return lr * sens.clamp_max(2.0) + rnd


def param_excess(got, ref, allowed):
    """max over the entries of |got - ref| / allowed (allowed: a number or a tensor of the parameter's shape)."""
    d = (torch.as_tensor(got).detach().cpu().double() - torch.as_tensor(ref).detach().cpu().double()).abs()
    return float((d / allowed).max()) if d.numel() else 0.0
