"""Self-consistency of the fused TRPL kernel, independent of the oracle (ITPAL is un-vendored, its parity unpinned, so
the build states its own tolerances): (1) the projected covariance meets the bound: |KL_cov(proj || old) - eps_cov| <= 2e-6 where the
bound was violated and the projection is the identity elsewhere; (2) the analytic gradients of the kernel agree with central finite
differences of the kernel's own loss sums -- through the implicit function eta(S) of the KL projection and through the closed forms
of the Frobenius / Wasserstein projections (for KL / W2 with trust_region_coeff = 0: their regression loss detaches the projection,
which a finite difference cannot; the Frobenius regression loss is differentiated through the projection and is included)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
B, A = 48, 6


def _inputs(seed):
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda:0")
    r = lambda *s: torch.randn(*s, generator=g)
    loc = (0.3 * r(B, A)).to(dev)
    sigma = (0.6 + 0.5 * torch.rand(B, A, generator=g)).to(dev)
    batch = {"loc": (loc.cpu() + 0.25 * r(B, A)).to(dev), "var": (0.5 + torch.rand(B, A, generator=g)).to(dev),
             "action": (loc.cpu() + 0.5 * r(B, A)).to(dev), "sample_log_prob": (-6.0 + 0.3 * r(B)).to(dev), "advantage": r(B).to(dev)}
    batch["loc"][0] = loc[0] + 1e-4          # inside the mean bound
    batch["var"][1] = sigma[1] ** 2 * 1.0005  # inside the covariance bound
    return loc, sigma, batch


def _loss(loc, sigma, batch, proj, coeff, ent):
    from geometry_rl_amd import ops
    sums, _, dloc, dsigma, _, pm, pv = ops.trpl_fwd_bwd(loc, sigma, batch, None, mean_bound=0.05, cov_bound=0.0025, trust_region_coeff=coeff,
                                                          entropy_coef=ent, critic_coef=0.0, clip_value=0.0, global_batch=B, adv_stats=None,
                                                          want_projection=True, proj_type=proj)
    s = sums.double().cpu()
    return float((s[0] + s[1] - ent * s[2]) / B), dloc, dsigma, pm, pv


def test_kl_projection_meets_the_bound():
    loc, sigma, batch = _inputs(0)
    _, _, _, pm, pv = _loss(loc, sigma, batch, 0, 1.0, 0.0)
    S, So, pS = (sigma.double() ** 2).cpu(), batch["var"].double().cpu(), pv.double().cpu()
    kl = lambda s_, o_: 0.5 * ((s_ / o_) ** 2 - 1.0 - 2.0 * (s_ / o_).log()).sum(-1)     # KL of N(., s^2) from N(., o^2), cov part
    before, after = kl(S, So), kl(pS, So)
    active = before > 0.0025
    assert active.sum() > 10 and (~active).sum() >= 1
    print(f"KL bound residual {float((after[active] - 0.0025).abs().max()):.2e}")
    assert float((after[active] - 0.0025).abs().max()) <= 2e-6
    assert torch.equal(pS[~active].float(), S[~active].float())
    # mean part: 1/2 maha <= bound everywhere after the projection
    mp = 0.5 * (((pm.double().cpu() - batch["loc"].double().cpu()) / So) ** 2).sum(-1)
    assert float(mp.max()) <= 0.05 * (1 + 1e-5)


@pytest.mark.parametrize("proj,coeff", [(0, 0.0), (2, 0.0), (1, 1.5)])
def test_analytic_gradients_match_finite_differences(proj, coeff):
    loc, sigma, batch = _inputs(1 + proj)
    ent = 0.01
    _, dloc, dsigma, _, _ = _loss(loc, sigma, batch, proj, coeff, ent)
    h = 2e-3
    worst = 0.0
    for which, base, grad in (("loc", loc, dloc), ("sigma", sigma, dsigma)):
        for b, i in ((2, 0), (5, 3), (17, 5), (30, 1), (41, 2)):
            vals = []
            for sgn in (+1.0, -1.0):
                t = base.clone()
                t[b, i] += sgn * h
                vals.append(_loss(t if which == "loc" else loc, t if which == "sigma" else sigma, batch, proj, coeff, ent)[0])
            fd = (vals[0] - vals[1]) / (float((base[b, i] + h) - (base[b, i] - h)))
            an = float(grad[b, i])
            worst = max(worst, abs(fd - an) / (abs(an) + 1e-4))
            assert abs(fd - an) <= 2e-3 * abs(an) + 2e-6, (which, b, i, fd, an)
    print(f"proj {proj}: worst relative FD mismatch {worst:.2e}")
