"""The FP32 chain kernel with lag products on the matrix cores (mcmc_amd/csrc/chain_lag_kernel.hpp): the pooled
lag products P_k, the head / tail sums and the ring carry against direct FP64 sums over the very samples the launch
wrote back (statistics.hpp:36-57 is evaluated from exactly these sums, mcmc_stats_finalize), and against the VALU form
of the same statistics (chain_kernel.hpp, MCMC_LAG_MFMA=0)."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu

R2WS_REF = (O.R2WS, 2.0, 0.54)


def _engine(dev, **kw):
    from mcmc_amd import api
    t = kw.pop("target")
    return api.ChainEngine(target=api.make_target(*t), device=dev, **kw)


@pytest.fixture
def lag_option():
    from mcmc_amd import lib as L

    def setopt(value):
        L.check(L.lib.mcmc_set_option(b"MCMC_LAG_MFMA", value))
    yield setopt
    L.lib.mcmc_set_option(b"MCMC_LAG_MFMA", None)


def _direct(s, center, n_lags, hist=None):
    """FP64 sums over samples s[t][chain] (+ history hist[k][chain] = y_{-1-k}): S1, S2, P_k, A_i, Z_j."""
    y = s.astype(np.float64) - center
    n = y.shape[0]
    full = y if hist is None else np.concatenate([hist[::-1].astype(np.float64), y], axis=0)
    off = 0 if hist is None else hist.shape[0]
    P = np.zeros(n_lags)
    for k in range(1, n_lags + 1):
        a = full[off:]                                   # y_t, t in the launch
        lo = off - k
        b = full[max(lo, 0):max(lo, 0) + n - max(-lo, 0)]  # y_{t-k} where it exists
        P[k - 1] = (a[max(-lo, 0):] * b).sum()
    return y.sum(), (y * y).sum(), P


@pytest.mark.parametrize("n_chains,n_steps,t0", [(1, 49, 0), (64, 50, 3), (63, 200, 8), (300, 1037, 12345), (1000, 4100, 1 << 33),
                                                 (257, 16 * 70, 16), (65, 1500, 15)])
@pytest.mark.parametrize("mode", ["cross_fast", "recip_precise"])
def test_lag_sums_equal_direct_sums(dev, n_chains, n_steps, t0, mode):
    """One window-starting launch: P_k = sum_chains sum_t y_t y_{t-k} to 2e-6 of sum |y_t y_{t-k}| (the (h, l) halves carry
    22 bits; FP32 accumulation over <= 1024 steps), head sums A_i and tail sums Z_j to FP32 rounding, S1 / S2 / accepts
    as the kernel without lags.  Launch lengths and start steps that are not multiples of the 16-step blocks."""
    from mcmc_amd import lib as L
    rule, math = {"cross_fast": (L.ACCEPT_CROSS, L.MATH_FAST), "recip_precise": (L.ACCEPT_RECIP, L.MATH_PRECISE)}[mode]
    n_lags, center = 49, 2.3
    eng = _engine(dev, n_chains=n_chains, n_coords=2, target=R2WS_REF, step=2.4, precision=32, accept_rule=rule, math_mode=math,
                  n_lags=n_lags, center=center, seed=99)
    eng.steps_done = t0
    samples = torch.empty((n_steps, 2, n_chains), dtype=torch.float32, device=dev)
    bits = torch.zeros((n_steps, 2, n_chains), dtype=torch.uint8, device=dev)
    st = eng.run(n_steps, samples=samples, accept_bits=bits).cpu().numpy()
    torch.cuda.synchronize()
    s, b = samples.cpu().numpy(), bits.cpu().numpy()
    for c in range(2):
        S1, S2, P = _direct(s[:, c, :], np.float32(center), n_lags)
        y = s[:, c, :].astype(np.float64) - np.float32(center)
        assert st[c, 0] == pytest.approx(S1, rel=1e-6, abs=1e-4 * n_chains)
        assert st[c, 1] == pytest.approx(S2, rel=2e-6)
        assert st[c, 2] == b[:, c, :].sum()
        for k in range(1, n_lags + 1):
            scale = np.abs(y[k:] * y[:-k]).sum() if n_steps > k else 0.0
            assert abs(st[c, 3 + k - 1] - P[k - 1]) <= 2e-6 * scale + 1e-9, (c, k, st[c, 3 + k - 1], P[k - 1])
        np.testing.assert_allclose(st[c, 3 + n_lags:3 + 2 * n_lags], y[:n_lags].sum(axis=1), rtol=1e-6, atol=1e-5 * n_chains)
        np.testing.assert_allclose(st[c, 3 + 2 * n_lags:], y[::-1][:n_lags].sum(axis=1), rtol=1e-6, atol=1e-5 * n_chains)


@pytest.mark.parametrize("cuts", [(100, 100), (49, 3, 150), (160, 7, 20, 1, 60), (64, 64, 64)])
def test_lag_ring_carry_over_split_launches(dev, cuts):
    """A window cut into launches (ring carried, window_start = 0 after the first; launches shorter than the 49 lags
    among them): the summed P_k equal the direct sums over the whole window, the last launch's tail sums are the
    window's, and the chains end where one launch takes them."""
    from mcmc_amd import lib as L
    n_chains, n_lags, center = 200, 49, 2.3
    kw = dict(n_chains=n_chains, n_coords=1, target=R2WS_REF, step=2.4, precision=32, accept_rule=L.ACCEPT_CROSS,
              math_mode=L.MATH_FAST, n_lags=n_lags, center=center, seed=5)
    n = sum(cuts)
    one = _engine(dev, **kw)
    one.steps_done = 21
    samples = torch.empty((n, 1, n_chains), dtype=torch.float32, device=dev)
    sa = one.run(n, samples=samples).cpu().numpy().copy()
    parts = _engine(dev, carry_ring=True, **kw)
    parts.steps_done = 21
    tot = np.zeros_like(sa)
    last = None
    for i, m in enumerate(cuts):
        last = parts.run(m, window_start=(i == 0)).cpu().numpy().copy()
        tot[:, :3 + n_lags] += last[:, :3 + n_lags]
        if i == 0:
            tot[:, 3 + n_lags:3 + 2 * n_lags] = last[:, 3 + n_lags:3 + 2 * n_lags]
    tot[:, 3 + 2 * n_lags:] = last[:, 3 + 2 * n_lags:]
    torch.cuda.synchronize()
    assert torch.equal(one.state, parts.state)
    s = samples.cpu().numpy()[:, 0, :]
    y = s.astype(np.float64) - np.float32(center)
    S1, S2, P = _direct(s, np.float32(center), n_lags)
    assert tot[0, 2] == sa[0, 2]
    for k in range(1, n_lags + 1):
        scale = np.abs(y[k:] * y[:-k]).sum()
        assert abs(tot[0, 3 + k - 1] - P[k - 1]) <= 2e-6 * scale, (k, tot[0, 3 + k - 1], P[k - 1])
        assert abs(sa[0, 3 + k - 1] - P[k - 1]) <= 2e-6 * scale
    np.testing.assert_allclose(tot[0, 3 + n_lags:], sa[0, 3 + n_lags:], rtol=1e-6, atol=1e-3)
    # the ring holds the exact last samples
    ring = parts.ring.cpu().numpy()[0]
    np.testing.assert_array_equal(ring, (s[::-1][:n_lags] - np.float32(center)).astype(np.float32))


def test_matrix_core_and_valu_lag_kernels_agree(dev, lag_option):
    """MCMC_LAG_MFMA=0 selects the VALU form of the lag products (chain_kernel.hpp): identical chains, S1 / S2 / accepts /
    head / tail sums to FP32 rounding, P_k within the two kernels' FP32 accumulation error; fewer lags than 49 too."""
    from mcmc_amd import lib as L
    out = {}
    for n_lags in (49, 16):
        for opt in (b"1", b"0"):
            lag_option(opt)
            eng = _engine(dev, n_chains=5000, n_coords=3, target=R2WS_REF, step=2.2, precision=32, accept_rule=L.ACCEPT_CROSS,
                          math_mode=L.MATH_FAST, n_lags=n_lags, center=2.3, seed=3)
            st = eng.run(2000).cpu().numpy().copy()
            torch.cuda.synchronize()
            out[opt] = (st, eng.state.cpu().numpy().copy(), eng.summary())
        (a, sa, ua), (b, sb, ub) = out[b"1"], out[b"0"]
        assert np.array_equal(sa, sb)
        np.testing.assert_array_equal(a[:, 2], b[:, 2])
        np.testing.assert_allclose(a[:, :2], b[:, :2], rtol=1e-6)
        np.testing.assert_allclose(a[:, 3 + n_lags:], b[:, 3 + n_lags:], rtol=1e-6, atol=1e-2)
        n = 5000 * 2000
        np.testing.assert_allclose(a[:, 3:3 + n_lags], b[:, 3:3 + n_lags], rtol=0, atol=3e-6 * n)
        for c in range(3):
            np.testing.assert_allclose(ua[c]["rho"], ub[c]["rho"], rtol=0, atol=5e-6)
            assert ua[c]["tau_int"] == pytest.approx(ub[c]["tau_int"], rel=1e-4)


def test_far_start_saturates_instead_of_poisoning(dev):
    """A chain started far outside the target (x0 = 1e6) leaves f16's range: its halves saturate, the statistics stay
    finite, and the other statistics are untouched."""
    from mcmc_amd import lib as L
    eng = _engine(dev, n_chains=128, n_coords=1, target=R2WS_REF, step=2.4, precision=32, accept_rule=L.ACCEPT_CROSS,
                  math_mode=L.MATH_FAST, n_lags=49, center=2.3, seed=1, x0=1.0e6)
    st = eng.run(100).cpu().numpy()
    torch.cuda.synchronize()
    assert np.all(np.isfinite(st))


def test_randomised_windows_agree_with_the_valu_kernel(dev, lag_option):
    """Random chain counts, lag counts (1 .. 49), targets, rules and cuts of a window into ring-carried launches (lengths
    from 1 step up, start steps anywhere in a 16-step block): the matrix-core kernel and the VALU kernel leave identical
    chains and acceptance counts, and their summed lag products / the window's head and tail sums agree to FP32 rounding."""
    from mcmc_amd import lib as L
    rng = np.random.default_rng(20251111)
    targets = [(O.R2WS, 2.0, 0.54), (O.WS, 3.0, 0.5), (O.GAUSS, 1.5, 0.0), (O.R2GAUSS, 0.8, 0.0)]
    rules = [(L.ACCEPT_CROSS, L.MATH_FAST), (L.ACCEPT_RECIP, L.MATH_FAST), (L.ACCEPT_LOG, L.MATH_FAST), (L.ACCEPT_NODIV, L.MATH_FAST),
             (L.ACCEPT_RECIP, L.MATH_PRECISE)]
    for case in range(12):
        n_chains = int(rng.choice([1, 63, 64, 65, 200, 257, 777]))
        n_lags = int(rng.choice([1, 2, 7, 16, 33, 49]))
        target = targets[case % 4]
        rule, math = rules[case % 5]
        first = n_lags + int(rng.integers(0, 70))          # a window's first launch holds at least n_lags steps
        cuts = [first] + [int(x) for x in rng.choice([1, 2, 5, 15, 16, 17, 48, 49, 50, 130], size=int(rng.integers(1, 5)))]
        t0 = int(rng.integers(0, 1000))
        res = {}
        for opt in (b"1", b"0"):
            lag_option(opt)
            eng = _engine(dev, n_chains=n_chains, n_coords=2, target=target, step=1.7, precision=32, accept_rule=rule,
                          math_mode=math, n_lags=n_lags, center=1.0, seed=1000 + case, carry_ring=True)
            eng.steps_done = t0
            tot = None
            for i, m in enumerate(cuts):
                last = eng.run(m, window_start=(i == 0)).cpu().numpy().copy()
                if tot is None:
                    tot = last.copy()
                else:
                    tot[:, :3 + n_lags] += last[:, :3 + n_lags]
                tot[:, 3 + 2 * n_lags:] = last[:, 3 + 2 * n_lags:]
            torch.cuda.synchronize()
            res[opt] = (tot, eng.state.cpu().numpy().copy(), eng.ring.cpu().numpy().copy())
        (a, sa, ra), (b, sb, rb) = res[b"1"], res[b"0"]
        tag = (case, n_chains, n_lags, cuts, t0)
        assert np.array_equal(sa, sb), tag
        assert np.array_equal(ra, rb), tag
        np.testing.assert_array_equal(a[:, 2], b[:, 2], err_msg=str(tag))
        n = n_chains * sum(cuts)
        scale = max(float(np.abs(b[:, 1]).max()), 1.0)      # sum y^2 bounds every sum |y_t y_{t-k}|
        np.testing.assert_allclose(a[:, :2], b[:, :2], rtol=2e-6, atol=1e-4 * n, err_msg=str(tag))
        np.testing.assert_allclose(a[:, 3:3 + n_lags], b[:, 3:3 + n_lags], rtol=0, atol=4e-6 * scale + 1e-6 * n, err_msg=str(tag))
        np.testing.assert_allclose(a[:, 3 + n_lags:], b[:, 3 + n_lags:], rtol=2e-6, atol=2e-5 * n_chains + 1e-4, err_msg=str(tag))
