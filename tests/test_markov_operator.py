"""The chain statistics the stepsize study reports -- acceptance, lag-k autocorrelation, integrated autocorrelation time
(stepsize/src/main.cpp:14-36, statistics.hpp:36-57, StepSizeAnalysis.ipynb cell 1) -- against values computed from the
TRANSITION OPERATOR of the random-walk Metropolis chain (tests/markov_reference.py): no samples, no restatement of the
reference's code, only the algorithm's definition.  The reference's own recorded end points (StepSizeAnalysis.ipynb:
acceptance 0.2280 at width 5.00) are reproduced by the operator, which ties the two together.
CPU: the operator's convergence and the recorded end point.  GPU: the streaming statistics of the chain kernels (49 lag
products on the matrix cores, and the VALU form) at 10^9 samples."""
import numpy as np
import pytest

import markov_reference as M

R2WS = ("r2ws", 2.0, 0.54)          # PDF(), metropolis.hpp:19-23


def test_operator_converges_and_reproduces_the_recorded_acceptance():
    p = M.density(*R2WS)
    ext, left = M.richardson(p, 2.48, 14.0, 1200)
    fine = M.chain_statistics(p, 2.48, 14.0, 2400)
    assert left < 1e-7                                              # rho_k: the discretisation error that is left
    assert abs(fine["tau_int"] - ext["tau_int"]) < 1e-6 and abs(fine["acceptance"] - ext["acceptance"]) < 2e-6
    assert ext["mean"] == pytest.approx(2.3317236, abs=2e-7)        # quadrature (profiles/r3_known_answer_kinds.jsonl)
    assert ext["variance"] == pytest.approx(6.424247 - 2.3317236 ** 2, abs=2e-6)   # <r^2> = 6.424247
    wide = M.chain_statistics(p, 5.0, 14.0, 2400)
    assert wide["acceptance"] == pytest.approx(0.2280, abs=5e-4)    # StepSizeAnalysis.ipynb's recorded value at width 5.00
    # Statistics of an exactly solvable chain: on a flat density every proposal inside [0, L] is accepted
    flat = M.chain_statistics(lambda r: np.ones_like(r), 0.5, 10.0, 2000)
    from scipy import stats
    x = (np.arange(2000) + 0.5) * (10.0 / 2000)
    inside = np.mean(stats.norm.cdf((10.0 - x) / 0.5) - stats.norm.cdf((0.0 - x) / 0.5))
    assert flat["acceptance"] == pytest.approx(inside, abs=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind, p0, p1, width, upper", [("r2ws", 2.0, 0.54, 2.48, 14.0), ("r2gauss", 1.0, 0.0, 1.6, 9.0)])
@pytest.mark.parametrize("lag_mfma", ["1", "0"])
def test_streaming_statistics_against_the_transition_operator(dev, kind, p0, p1, width, upper, lag_mfma):
    import torch
    from mcmc_amd import api, lib as L
    want, left = M.richardson(M.density(kind, p0, p1), width, upper, 1500)
    assert left < 1e-6
    code = {"r2ws": L.TARGET_R2WS, "r2gauss": L.TARGET_R2GAUSS}[kind]
    n_chains, n_steps = 1 << 18, 4096
    L.check(L.lib.mcmc_set_option(b"MCMC_LAG_MFMA", lag_mfma.encode()))
    try:
        eng = api.ChainEngine(n_chains, 1, api.make_target(code, p0, p1), width, precision=32, accept_rule=L.ACCEPT_CROSS,
                              math_mode=L.MATH_FAST, n_lags=L.MAX_LAGS, seed=20261112, x0=1.0, center=want["mean"], device=dev)
        eng.run(512, n_lags=0)
        eng.run(n_steps, window_start=True, n_lags=L.MAX_LAGS)
        torch.cuda.synchronize()
        got = eng.summary()[0]
    finally:
        L.lib.mcmc_set_option(b"MCMC_LAG_MFMA", None)
    n = n_chains * n_steps
    tau = want["tau_int"]
    se_mean = np.sqrt(want["variance"] * tau / n)
    assert abs(got["mean"] - want["mean"]) < 4 * se_mean
    assert got["variance"] == pytest.approx(want["variance"], rel=4 * np.sqrt(2 * tau / n) + 1e-5)
    assert got["acceptance"] == pytest.approx(want["acceptance"], abs=4 * np.sqrt(0.25 * tau / n))
    # rho_k: standard error ~ sqrt(tau_int / n) (Bartlett); 5 of them over 49 lags
    np.testing.assert_allclose(got["rho"], want["rho"], rtol=0, atol=5 * np.sqrt(tau / n) + 2e-5)
    assert got["tau_int"] == pytest.approx(tau, rel=4e-3)
