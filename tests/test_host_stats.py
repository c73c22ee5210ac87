"""Host-side logic of the C ABI that needs no GPU: the statistics functions with
the reference's arithmetic, finalisation of the streaming sufficient statistics
(the formulas the device accumulators feed), and the sigma integration --
against the reference-generated golden vectors and the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _lib():
    from mcmc_amd import lib as L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def test_f64_statistics_match_reference_golden():
    """mcmc_Average_f64 / Variance_f64 / AutoCorrelation_f64 == statistics.hpp:6-57 bit for bit."""
    L = _lib()
    g = np.load(os.path.join(GOLD, "statistics_a4.npz"))
    x = np.ascontiguousarray(g["series"])
    n = len(x)
    assert L.lib.mcmc_Average_f64(_dp(x), n) == g["ref_average"][0]
    assert L.lib.mcmc_Variance_f64(_dp(x), n) == g["ref_variance"][0]
    for k in range(1, 50):
        assert L.lib.mcmc_AutoCorrelation_f64(_dp(x), n, k) == g["ref_autocorrelation"][k - 1]


def test_f32_statistics_match_oracle(oracle):
    """mcmc_Sum / Average / Variance / AutoCorrelation == Statistics.cuh:7-87 restatement bit for bit."""
    L = _lib()
    rng = np.random.default_rng(3)
    for n in (1, 2, 17, 32768):
        x = (6.4 + rng.standard_normal(n)).astype(np.float32)
        assert L.lib.mcmc_Sum(_fp(x), n) == oracle.sum_f32(x)
        avg = L.lib.mcmc_Average(_fp(x), n)
        assert avg == oracle.average_f32(x)
        assert L.lib.mcmc_Variance(_fp(x), C.c_float(avg), n) == oracle.variance_f32(x, avg)
        if n > 3:
            for k in (1, 3):
                assert L.lib.mcmc_AutoCorrelation(_fp(x), C.c_float(avg), k, n) == oracle.autocorrelation_f32(x, avg, k)


def test_f32_statistics_overloads_that_average_themselves(oracle):
    """mcmc_Variance_n / mcmc_AutoCorrelation_n == Statistics.cuh:45-60 / :90-115 (restated with their own loops
    in the oracle), bit for bit; and equal to the avg-taking forms fed the Kahan average."""
    L = _lib()
    rng = np.random.default_rng(5)
    for n in (1, 2, 17, 4096, 32768):
        x = (6.4 + rng.standard_normal(n)).astype(np.float32)
        v = L.lib.mcmc_Variance_n(_fp(x), n)
        assert v == oracle.variance_n_f32(x)
        avg = L.lib.mcmc_Average(_fp(x), n)
        assert v == L.lib.mcmc_Variance(_fp(x), C.c_float(avg), n)
        if n > 3:
            for k in (1, 2, 7):
                r = L.lib.mcmc_AutoCorrelation_n(_fp(x), k, n)
                assert r == oracle.autocorrelation_n_f32(x, k)
                assert r == L.lib.mcmc_AutoCorrelation(_fp(x), C.c_float(avg), k, n)
    # AR(1) series: the lag-1 value is what it should be
    y = np.zeros(20000, dtype=np.float32)
    e = rng.standard_normal(20000).astype(np.float32)
    for i in range(1, len(y)):
        y[i] = 0.8 * y[i - 1] + e[i]
    assert abs(L.lib.mcmc_AutoCorrelation_n(_fp(y), 1, len(y)) - 0.8) < 0.02


def test_run_estimator_matches_oracle(oracle):
    """main.cu:253-258: mean of run means, stderr = sqrt(popvar/R), RMS, RMS_stderr."""
    from mcmc_amd import api
    rng = np.random.default_rng(4)
    runs = (6.42 + 0.01 * rng.standard_normal(16)).astype(np.float32)
    m, se, rms, rmse = oracle.run_estimator_f32(runs)
    avg = api.Average(runs)
    assert avg == m
    var = api.Variance(runs, avg)
    assert np.float32(np.sqrt(np.float32(var / np.float32(16)))) == np.float32(se)
    assert np.float32(np.sqrt(np.float32(avg))) == np.float32(rms)


def _suff_stats(s, n_lags, center):
    """[n_steps][n_chains] samples -> the vector the device accumulates (mcmc_hip.h MCMC_STATS_LEN)."""
    y = s - center
    n = y.shape[0]
    out = np.zeros(3 + 3 * n_lags)
    out[0], out[1] = y.sum(), (y * y).sum()
    for k in range(1, n_lags + 1):
        out[3 + k - 1] = (y[:-k] * y[k:]).sum()
        out[3 + n_lags + k - 1] = y[k - 1].sum()           # A_{k-1}: k-th sample of the window
        out[3 + 2 * n_lags + k - 1] = y[n - k].sum()       # Z_{k-1}: k-th last sample
    return out


@pytest.mark.parametrize("center", [0.0, 2.3])
def test_stats_finalize_reproduces_reference_autocorrelation(center):
    """One chain: finalising the streaming sums gives statistics.hpp's Average / Variance / AutoCorrelation."""
    from mcmc_amd import api
    g = np.load(os.path.join(GOLD, "statistics_a4.npz"))
    x = g["series"]
    st = _suff_stats(x[:, None], 49, center)
    st[2] = 1234
    out = api.finalize_stats(st, 49, 1, len(x), center)
    assert out["mean"] == pytest.approx(g["ref_average"][0], rel=1e-13)
    assert out["variance"] == pytest.approx(g["ref_variance"][0], rel=1e-10)
    np.testing.assert_allclose(out["rho"], g["ref_autocorrelation"], rtol=0, atol=1e-10)
    assert out["acceptance"] == pytest.approx(1234 / len(x))
    assert out["tau_int"] == pytest.approx(1 + 2 * g["ref_autocorrelation"].sum(), rel=1e-9)


def test_stats_finalize_pooled_chains_and_additivity():
    """Many chains pooled, and the sum of two half-partitions' vectors (what the all-reduce does)."""
    from mcmc_amd import api
    rng = np.random.default_rng(11)
    n, nc, L = 400, 6, 12
    s = np.cumsum(rng.standard_normal((n, nc)), axis=0) * 0.05 + 2.0
    mu = s.mean()
    var = ((s - mu) ** 2).sum() / (s.size - 1)
    whole = _suff_stats(s, L, 2.0)
    parts = _suff_stats(s[:, :3], L, 2.0) + _suff_stats(s[:, 3:], L, 2.0)
    np.testing.assert_allclose(parts, whole, rtol=1e-12, atol=1e-12)
    out = api.finalize_stats(parts, L, nc, n, 2.0)
    assert out["mean"] == pytest.approx(mu, rel=1e-12)
    assert out["variance"] == pytest.approx(var, rel=1e-9)
    for k in range(1, L + 1):
        cov = ((s[:-k] - mu) * (s[k:] - mu)).sum() / (nc * (n - k) - 1)
        assert out["rho"][k - 1] == pytest.approx(cov / var, abs=1e-9)


def test_cross_sections_match_reference_python_golden():
    """mcmc_cross_sections == xsection/tools/XSectionAnalysis.py:3-28 (values generated by importing it)."""
    from mcmc_amd import api
    g = np.load(os.path.join(GOLD, "sigma_f1.npz"))
    cs = api.CrossSections(g["b"], g["S_AB"], g["S_AB_stderr"])
    got = [cs["tot"][0], cs["tot"][1], cs["el"][0], cs["el"][1], cs["rxn"][0], cs["rxn"][1]]
    np.testing.assert_allclose(got, g["ref_sigma"], rtol=1e-12)
    with pytest.raises(_lib().McmcError):
        api.CrossSections(g["b"][:1], g["S_AB"][:1], g["S_AB_stderr"][:1])
