"""S_AB against answers that depend on no restatement: for R2Gaussian radial densities (Distributions.cuh:25-28) with
isotropic angles (IntegrateSAB.cuh:50-61) a nucleon's position is a 3-D Gaussian N(0, v I), so its transverse position
is N(0, v I_2) and the profile average is analytic.  With Gamma(d^2) = sigma / (4 pi beta) exp(-d^2 / (2 beta))
(IntegrateSAB.cuh:9-17):

  one nucleon on one nucleon          S(b) = 1 - sigma / (4 pi (beta + 2 v)) exp(-b^2 / (2 (beta + 2 v)))
  one nucleon on n (the product over pairs that share the A nucleon, IntegrateSAB.cuh:63-78)
                                      S(b) = E_{s ~ N(0, v I_2)} [ (1 - g(b x + s))^n ],
                                      g(c) = sigma / (4 pi (beta + v)) exp(-|c|^2 / (2 (beta + v)))   (2-D quadrature)

This pins the radial chain, the angle transformation, Gamma, the pair product and the Kahan mean together, at
estimate level (the bar of BASELINE.json: 3 combined standard errors; 4 here over 6-8 comparisons per test).
CPU: the oracle's MCIntegrate_S_AB restatement on numpy draws.  GPU: the S_AB kernels on the library's Philox draws -- four
lanes per event in two math modes, and 64 lanes per event (one nucleon on 200) with the exact pair culling, where the
black-disk inelastic fraction and the mean number of colliding pairs have analytic values as well."""
import numpy as np
import pytest
from scipy import integrate

import oracle_lib as O

SIGMA, BETA = 4.3, 0.2


def s_one_on_one(b, v):
    w = BETA + 2.0 * v
    return 1.0 - SIGMA / (4.0 * np.pi * w) * np.exp(-b * b / (2.0 * w))


def s_one_on_n(b, v, n):
    w = BETA + v
    g = lambda x, y: SIGMA / (4.0 * np.pi * w) * np.exp(-((b + x) ** 2 + y * y) / (2.0 * w))  # noqa: E731
    f = lambda y, x: np.exp(-(x * x + y * y) / (2.0 * v)) / (2.0 * np.pi * v) * (1.0 - g(x, y)) ** n  # noqa: E731
    r = 9.0 * np.sqrt(v)
    return integrate.dblquad(f, -r, r, -r, r, epsabs=1e-10, epsrel=1e-10)[0]


def test_closed_forms_agree_with_each_other():
    for b in (0.0, 1.1):
        assert s_one_on_n(b, 1.3, 1) == pytest.approx(s_one_on_one(b, 1.3), abs=1e-8)


@pytest.mark.parametrize("nB, v", [(1, 1.3), (3, 0.8)])
def test_oracle_sab_against_the_analytic_profile_average(oracle, nB, v):
    """orc_integrate_sab_thread (IntegrateSAB.cuh:19-97 restated) on numpy draws: threads started IN the stationary
    distribution (|N(0, v I_3)|), so there is no warm-up bias to wait out."""
    rng = np.random.default_rng(20261112 + nB)
    nN, threads, samples = 1 + nB, 192, 1536
    tg = [(O.R2GAUSS, v, 0.0)] * nN
    for b in (0.0, 1.2, 2.5):
        means = np.empty(threads)
        for e in range(threads):
            d = np.empty((samples, 2, nN, 2), dtype=np.float32)
            d[:, 0, :, 0] = rng.standard_normal((samples, nN))
            d[:, 0, :, 1] = 1.0 - rng.random((samples, nN))            # curand_uniform is (0, 1]
            d[:, 1] = 1.0 - rng.random((samples, nN, 2))
            x0 = np.linalg.norm(rng.standard_normal((nN, 3)), axis=1) * np.sqrt(v)
            means[e] = oracle.integrate_sab_thread(1, nB, tg, x0, 1.8 * np.sqrt(v), samples, b, d.reshape(-1),
                                                   sigmaNN=SIGMA, beta=BETA)[1]
        want = s_one_on_one(b, v) if nB == 1 else s_one_on_n(b, v, nB)
        err = means.std(ddof=1) / np.sqrt(threads)
        assert err < 1.5e-3 and abs(means.mean() - want) < 4 * err, (nB, b, means.mean(), want, err)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["cross_fast", "recip_precise"])
@pytest.mark.parametrize("nB, v", [(1, 1.3), (3, 0.8)])
def test_device_sab_against_the_analytic_profile_average(dev, mode, nB, v):
    import torch
    from mcmc_amd import api, lib as L
    rule, math = {"cross_fast": (L.ACCEPT_CROSS, L.MATH_FAST), "recip_precise": (L.ACCEPT_RECIP, L.MATH_PRECISE)}[mode]
    T, samples = 1 << 15, 128
    xs = api.XsEngine(1, nB, T, [(L.TARGET_R2GAUSS, v, 0.0)] * (1 + nB), stepsize=1.8 * np.sqrt(v), accept_rule=rule,
                      math_mode=math, sigmaNN=SIGMA, beta=BETA, seed=20261112, device=dev)
    xs.InitSampleFromPdf()
    xs.WarmupMetropolis(256)
    for b in (0.0, 0.6, 1.2, 2.5):
        res, mean, _ = xs.MCIntegrate_S_AB(samples, b)
        torch.cuda.synchronize()
        r = res.double().cpu().numpy()
        want = s_one_on_one(b, v) if nB == 1 else s_one_on_n(b, v, nB)
        err = r.std(ddof=1) / np.sqrt(T)
        assert float(mean[0]) == pytest.approx(r.mean(), rel=1e-6)          # the device row mean is the mean of the events
        assert err < 4e-4 and abs(r.mean() - want) < 4 * err, (mode, nB, b, r.mean(), want, err)


@pytest.mark.gpu
def test_general_kernel_at_the_launch_size_where_light_nuclei_change_kernels(dev):
    """From 2^17 events on, r^2 Woods-Saxon light nuclei take sab_small_kernel (one event per lane); Gaussian densities stay
    on the general kernel, so this is the general kernel's four-lanes-per-event path at that launch size against the
    one-on-one formula (the register kernel itself is compared with the general one in test_sab_light_nuclei_*)."""
    import torch
    from mcmc_amd import api, lib as L
    v, T = 1.3, 1 << 17
    xs = api.XsEngine(1, 1, T, [(L.TARGET_R2GAUSS, v, 0.0)] * 2, stepsize=2.0, accept_rule=L.ACCEPT_NODIV, math_mode=L.MATH_FAST,
                      sigmaNN=SIGMA, beta=BETA, seed=5, device=dev)
    xs.InitSampleFromPdf()
    xs.WarmupMetropolis(128)
    for b in (0.0, 1.5):
        res, _, _ = xs.MCIntegrate_S_AB(64, b)
        torch.cuda.synchronize()
        r = res.double().cpu().numpy()
        err = r.std(ddof=1) / np.sqrt(T)
        assert abs(r.mean() - s_one_on_one(b, v)) < 4 * err and err < 2e-4, (b, r.mean(), s_one_on_one(b, v), err)


@pytest.mark.gpu
@pytest.mark.parametrize("with_hits", [False, True])
def test_wavefront_per_event_kernel_with_pair_culling_against_the_analytic_profile_average(dev, with_hits):
    """One nucleon on 200 (v = 25 fm^2: a nucleus-sized cloud): 64 lanes per event, LDS pair tiles, the exact pair culling
    (most of the 200 are farther than the culling distance from the one A nucleon) -- product only, and product + black-disk
    counters, whose inelastic fraction has an analytic value too: a black-disk hit needs |b + s_A - s_j| < d_max for some j,
    P(no hit) = E_s[(1 - q(b x + s))^n] with q(c) = P(|c - s_j| < d_max), s_j ~ N(0, v I_2): a non-central chi-square CDF."""
    import torch
    from scipy import stats
    from mcmc_amd import api, lib as L
    v, n, T, samples = 25.0, 200, 1 << 13, 64
    dmax2 = SIGMA / np.pi
    xs = api.XsEngine(1, n, T, [(L.TARGET_R2GAUSS, v, 0.0)] * (1 + n), stepsize=1.8 * np.sqrt(v), accept_rule=L.ACCEPT_CROSS,
                      math_mode=L.MATH_FAST, sigmaNN=SIGMA, beta=BETA, seed=99, device=dev)
    xs.InitSampleFromPdf()
    xs.WarmupMetropolis(128)

    def p_no_hit(b):
        q = lambda x, y: stats.ncx2.cdf(dmax2 / v, 2, ((b + x) ** 2 + y * y) / v)  # noqa: E731
        f = lambda y, x: np.exp(-(x * x + y * y) / (2.0 * v)) / (2.0 * np.pi * v) * (1.0 - q(x, y)) ** n  # noqa: E731
        r = 8.0 * np.sqrt(v)
        return integrate.dblquad(f, -r, r, -r, r, epsabs=1e-8, epsrel=1e-8)[0]

    for b in (0.0, 4.0, 9.0):
        res, _, hits = xs.MCIntegrate_S_AB(samples, b, dmax2=dmax2 if with_hits else None)
        torch.cuda.synchronize()
        r = res.double().cpu().numpy()
        err = r.std(ddof=1) / np.sqrt(T)
        want = s_one_on_n(b, v, n)
        assert err < 2e-3 and abs(r.mean() - want) < 4 * err, (with_hits, b, r.mean(), want, err)
        if with_hits:
            h = hits.cpu().numpy()          # colliding pairs, samples with >= 1 collision, participants, samples
            assert h[3] == T * samples
            p_inel, want_inel = h[1] / h[3], 1.0 - p_no_hit(b)
            # event samples of one chain are correlated (tau ~ 5 sweeps): binomial error inflated by sqrt(2 tau)
            err_inel = np.sqrt(want_inel * (1 - want_inel) / h[3] * 10.0)
            assert abs(p_inel - want_inel) < 4 * err_inel, (b, p_inel, want_inel, err_inel)
            # mean number of colliding pairs = n E_s[q(b x + s_A)] with s_A - s_j ~ N(0, 2 v I_2)
            ncoll_want = n * stats.ncx2.cdf(dmax2 / (2 * v), 2, b * b / (2 * v))
            assert h[0] / h[3] == pytest.approx(ncoll_want, rel=0.02), (b, h[0] / h[3], ncoll_want)
            # the inelastic-flag kernel (lazy angles, probes, early exit: mcmc_launch_integrate_inel) on fresh samples
            hi = xs.MCIntegrate_Inel(samples, b, dmax2).cpu().numpy()
            assert hi[1] == T * samples and abs(hi[0] / hi[1] - want_inel) < 4 * err_inel, (b, hi[0] / hi[1], want_inel)
