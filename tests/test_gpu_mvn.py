"""GPU parity tests of the correlated-Gaussian "stress" path (mcmc_mvn_run, BASELINE.json
configs[4]) against the CPU oracle's restatement of the same sweep, through the C ABI.
The reference has no such target: the sweep is its GibbsStep / MetropolisHastingsStep
(Metropolis.cuh:40-79) on the conditional densities with log-space acceptance, so parity
is "unpinned" against reference outputs and pinned to the oracle + closed-form answers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(dev, n_chains, dim, rho, step, lag, **kw):
    from mcmc_amd import api
    qd, qo = api.ar1_precision(dim, rho)
    return api.MvnEngine(n_chains, qd, qo, step, lag, device=dev, **kw), qd, qo


@pytest.mark.parametrize("mode", ["precise", "fast"])
@pytest.mark.parametrize("dim", [8, 64])
def test_mvn_injected_matches_oracle(dev, oracle, mode, dim):
    from mcmc_amd import lib as L
    n_chains, sweeps, step = 6, 40, 0.9
    rng = np.random.default_rng(dim)
    inj = np.empty((sweeps, dim, 2, n_chains), dtype=np.float32)
    inj[:, :, 0] = rng.standard_normal((sweeps, dim, n_chains))
    inj[:, :, 1] = 1.0 - rng.random((sweeps, dim, n_chains))
    eng, qd, qo = _engine(dev, n_chains, dim, 0.9, step, lag=3,
                          math_mode=L.MATH_PRECISE if mode == "precise" else L.MATH_FAST)
    full = torch.empty((sweeps, dim, n_chains), dtype=torch.float32, device=dev)
    bits = torch.zeros((sweeps, dim, n_chains), dtype=torch.uint8, device=dev)
    obs = torch.empty((sweeps, n_chains), dtype=torch.float32, device=dev)
    eng.run(sweeps, samples_obs=obs, samples_full=full, inj=torch.from_numpy(inj).to(dev), accept_bits=bits)
    torch.cuda.synchronize()
    f, b, o = full.cpu().numpy(), bits.cpu().numpy(), obs.cpu().numpy()
    strict = 0
    for c in range(n_chains):
        d = np.ascontiguousarray(inj[:, :, :, c])
        acc, x, trace, rb, margin = oracle.mvn_chain(qd, qo, [1.0] * dim, step, sweeps, draws=d)
        if margin > 1e-5:
            assert np.array_equal(b[:, :, c], rb), f"accept bits differ for chain {c}"
            np.testing.assert_allclose(f[:, :, c], trace, rtol=1e-6, atol=1e-7)
            strict += 1
    assert strict >= n_chains - 2
    assert np.array_equal(o, f[:, 0, :])                       # observable = coordinate 0
    assert np.array_equal(eng.state.cpu().numpy(), f[-1])      # final state = last sample


def test_mvn_philox_matches_oracle_stream(dev, oracle):
    """Philox mode: the device's packed draw stream (tag 3) and sweep == the oracle's, for launches that
    start inside a group of 8 proposals' worth of sweeps (offset carried by steps_done)."""
    from mcmc_amd import lib as L
    dim, n_chains, step, seed = 8, 70, 1.1, 77
    eng, qd, qo = _engine(dev, n_chains, dim, 0.9, step, lag=2, math_mode=L.MATH_PRECISE, seed=seed, chain_offset=1000)
    f1 = torch.empty((5, dim, n_chains), dtype=torch.float32, device=dev)
    f2 = torch.empty((30, dim, n_chains), dtype=torch.float32, device=dev)
    eng.run(5, samples_full=f1)
    eng.run(30, samples_full=f2)
    torch.cuda.synchronize()
    f = np.concatenate([f1.cpu().numpy(), f2.cpu().numpy()])
    checked = 0
    for c in (0, 1, 63, 64, 69):
        _, _, trace, _, margin = oracle.mvn_chain(qd, qo, [1.0] * dim, step, 35, seed=seed, chain=1000 + c)
        if margin > 1e-5:
            np.testing.assert_allclose(f[:, :, c], trace, rtol=2e-6, atol=1e-6)
            checked += 1
    assert checked >= 3


@pytest.mark.parametrize("lag", [1, 5, 16])
def test_mvn_lag_ring_statistics(dev, lag):
    """Lag-L autocorrelation through the HBM ring == statistics.hpp:36-57 evaluated on the written-back
    series (pooled over chains), with the window split over three launches of unequal length."""
    dim, n_chains, step = 8, 333, 1.0
    eng, _, _ = _engine(dev, n_chains, dim, 0.9, step, lag, center=0.25, observable=3)
    eng.run(50)                                       # warm-up (its statistics are discarded)
    lens = (max(lag, 7), 41, 64)
    n = sum(lens)
    w = (eng.steps_done, eng.steps_done + n)
    tot = torch.zeros(6, dtype=torch.float64, device=dev)
    series = []
    for k in lens:
        obs = torch.empty((k, n_chains), dtype=torch.float32, device=dev)
        tot += eng.run(k, window=w, samples_obs=obs)
        series.append(obs)
    torch.cuda.synchronize()
    s = torch.cat(series).cpu().numpy().astype(np.float64)
    summ = eng.summary(tot.cpu().numpy(), n)
    mu = s.mean()
    var = ((s - mu) ** 2).sum() / (s.size - 1)
    cov = ((s[:-lag] - mu) * (s[lag:] - mu)).sum() / (n_chains * (n - lag) - 1)
    assert summ["mean"] == pytest.approx(mu, abs=1e-6)
    assert summ["variance"] == pytest.approx(var, rel=1e-5)
    assert summ["rho_lag"] == pytest.approx(cov / var, abs=2e-5)
    assert 0.2 < summ["acceptance"] < 0.9


def test_mvn_known_answer_and_partition(dev):
    """2^16 chains x 64 dims, Sigma_ij = 0.9^|i-j|: every coordinate has mean 0 and variance 1, neighbours
    correlate at 0.9; and two half partitions with chain_offset reproduce the single launch exactly."""
    from mcmc_amd import lib as L
    dim, N, step = 64, 1 << 16, 0.8
    eng, _, _ = _engine(dev, N, dim, 0.9, step, lag=4, x0=0.0)
    eng.run(1500)
    st = eng.run(100).clone()
    torch.cuda.synchronize()
    x = eng.state.cpu().numpy().astype(np.float64)
    se = 1.0 / np.sqrt(N)
    assert np.all(np.abs(x.mean(axis=1)) < 5 * se)
    assert np.all(np.abs(x.var(axis=1) - 1.0) < 5 * np.sqrt(2.0) * se)
    c01 = np.corrcoef(x[10], x[11])[0, 1]
    assert c01 == pytest.approx(0.9, abs=0.01)
    summ = eng.summary(st.cpu().numpy(), 100)
    assert summ["variance"] == pytest.approx(1.0, abs=0.02) and abs(summ["mean"]) < 0.02
    # partition invariance (the multi-GPU sharding)
    a, _, _ = _engine(dev, 1000, 8, 0.9, 1.0, lag=3, seed=9)
    h0, _, _ = _engine(dev, 500, 8, 0.9, 1.0, lag=3, seed=9, chain_offset=0)
    h1, _, _ = _engine(dev, 500, 8, 0.9, 1.0, lag=3, seed=9, chain_offset=500)
    sa, s0, s1 = a.run(64).clone(), h0.run(64).clone(), h1.run(64).clone()
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([h0.state, h1.state], dim=1), a.state)
    np.testing.assert_allclose((s0 + s1).cpu().numpy(), sa.cpu().numpy(), rtol=1e-12, atol=1e-9)


def test_mvn_argument_errors(dev):
    from mcmc_amd import lib as L
    eng, _, _ = _engine(dev, 10, 8, 0.5, 1.0, lag=4)
    with pytest.raises(L.McmcError):
        eng.run(2)                       # window shorter than the lag
    eng.args.dim = 12
    with pytest.raises(L.McmcError):
        eng.run(10)


def test_mvn_stress_configuration_full_size(dev):
    """BASELINE configs[4] at its per-GPU size: 2^21 chains x 64 dimensions, lag 1000 (8.4 GB ring), observable
    written back every sweep.  Size-independent properties, all evaluated on the device: the lag-1000
    autocorrelation from the ring accumulators == the one computed from the written-back series
    (statistics.hpp:36-57, pooled over chains), mean/variance likewise, and the window split over two launches
    gives the same state as one launch."""
    N, dim, lag, n = 1 << 21, 64, 1000, 1100
    eng, _, _ = _engine(dev, N, dim, 0.9, 0.8, lag, x0=0.0, seed=5)
    ref, _, _ = _engine(dev, N, dim, 0.9, 0.8, lag, x0=0.0, seed=5)
    eng.run(64, window=(0, 64 + lag))   # warm-up: its statistics are discarded (any window >= lag sweeps will do)
    ref.run(64, window=(0, 64 + lag))
    w = (eng.steps_done, eng.steps_done + n)
    obs = torch.empty((n, N), dtype=torch.float32, device=dev)
    tot = eng.run(700, window=w, samples_obs=obs[:700]).clone()
    tot += eng.run(400, window=w, samples_obs=obs[700:])
    ref.run(n, window=w)
    torch.cuda.synchronize()
    assert torch.equal(eng.state, ref.state)
    summ = eng.summary(tot.cpu().numpy(), n)
    # statistics.hpp on the device, in FP64, chunked over chains
    s1 = s2 = 0.0
    for c in range(0, N, 1 << 18):
        blk = obs[:, c:c + (1 << 18)].double()
        s1 += blk.sum().item()
        s2 += (blk * blk).sum().item()
    cnt = float(n) * N
    mu = s1 / cnt
    var = (s2 - cnt * mu * mu) / (cnt - 1)
    cov = 0.0
    for c in range(0, N, 1 << 18):
        blk = obs[:, c:c + (1 << 18)].double() - mu
        cov += (blk[:-lag] * blk[lag:]).sum().item()
    cov /= (float(n - lag) * N - 1)
    assert summ["mean"] == pytest.approx(mu, abs=1e-6)
    assert summ["variance"] == pytest.approx(var, rel=1e-5)
    assert summ["rho_lag"] == pytest.approx(cov / var, abs=2e-5)
    assert abs(summ["rho_lag"]) < 0.02          # lag 1000 >> tau_int of one coordinate of the AR(1) target
    assert 0.2 < summ["acceptance"] < 0.9


@pytest.mark.parametrize("mode", ["fast", "precise"])
def test_mvn_register_kernel_equals_lds_kernel(dev, mode):
    """The stress configuration's own shape (64-D, observable 0, Philox draws, observable write-back only) takes a
    kernel with the chain's coordinates in registers.  Same draws, same expressions in the same order as the LDS
    kernel: states, the six statistics, the observable series and the lag ring are bit-identical -- for a ragged chain
    count over several tiles, two launches of one window, and a launch that starts mid-way through a draw group."""
    from mcmc_amd import lib as L
    dim, N, lag = 64, 2048 * 3 + 77, 5
    out = []
    try:
        for reg in ("1", "0"):
            L.check(L.lib.mcmc_set_option(b"MCMC_MVN_REG", reg.encode()))
            eng, _, _ = _engine(dev, N, dim, 0.9, 0.8, lag, x0=0.3, seed=21, chain_offset=555, center=0.1,
                                math_mode=L.MATH_FAST if mode == "fast" else L.MATH_PRECISE)
            eng.run(3, window=(0, 3 + lag))
            w = (eng.steps_done, eng.steps_done + 40)
            obs = torch.empty((40, N), dtype=torch.float32, device=dev)
            s1 = eng.run(13, window=w, samples_obs=obs[:13]).clone()
            s2 = eng.run(27, window=w, samples_obs=obs[13:]).clone()
            # and with every sweep's full state written back (the register kernel's FULL variant)
            full = torch.full((9, dim, N), 7.0, dtype=torch.float32, device=dev)
            s3 = eng.run(9, window=(eng.steps_done, eng.steps_done + 9), samples_full=full).clone()
            torch.cuda.synchronize()
            assert torch.equal(full[-1], eng.state)
            out.append((eng.state.clone(), eng.ring.clone(), obs.clone(), s1, s2, full.clone(), s3))
    finally:
        L.lib.mcmc_set_option(b"MCMC_MVN_REG", None)
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    assert out[0][3][2].item() > 0 and out[0][2].abs().max().item() > 0
