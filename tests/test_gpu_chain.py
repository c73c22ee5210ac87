"""GPU parity tests of the many-chain MH kernel (mcmc_chain_run) against the CPU
oracle, through the C ABI.  Bars (BASELINE.json north_star): under identical
injected draws accept/reject bits are identical and positions agree to 1e-6
relative; estimates agree with known answers within 3 combined standard errors.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu

R2WS_REF = (O.R2WS, 2.0, 0.54)      # PDF(), stepsize/include/metropolis.hpp:19-23
R2GAUSS_1 = (O.R2GAUSS, 1.0, 0.0)   # metropolis.hpp:24-25 (commented alternative), BASELINE config 0


def _engine(dev, **kw):
    from mcmc_amd import api, lib as L
    tk, p0, p1 = kw.pop("target")
    return api.ChainEngine(target=api.make_target(tk, p0, p1), device=dev, **kw)


@pytest.mark.parametrize("target", [R2WS_REF, R2GAUSS_1])
@pytest.mark.parametrize("step", [0.5, 2.5, 5.0])
def test_fp64_cpu_rule_injected_matches_a1(dev, oracle, target, step):
    """a1: metropolis.hpp:28-50 with injected draws; FP64, short-circuit rule,
    uniforms consumed through a per-chain cursor."""
    from mcmc_amd import lib as L
    n_steps, n_chains = 4096, 5
    rng = np.random.default_rng(1234)
    normals = rng.standard_normal((n_steps, 1, n_chains))
    uniforms = rng.random((n_steps, 1, n_chains))
    eng = _engine(dev, n_chains=n_chains, n_coords=1, target=target, step=step, precision=64,
                  accept_rule=L.ACCEPT_SHORTCIRCUIT, math_mode=L.MATH_PRECISE)
    samples = torch.empty((n_steps, 1, n_chains), dtype=torch.float64, device=dev)
    bits = torch.zeros((n_steps, 1, n_chains), dtype=torch.uint8, device=dev)
    eng.run(n_steps, samples=samples, inj_normals=torch.from_numpy(normals).to(dev),
            inj_uniforms=torch.from_numpy(uniforms).to(dev), accept_bits=bits)
    torch.cuda.synchronize()
    s, b = samples.cpu().numpy(), bits.cpu().numpy()
    summ = eng.summary()
    tot_acc = 0
    for c in range(n_chains):
        acc, ref_s, ref_b, used = oracle.mh_chain_f64(n_steps, step, target[0], target[1], target[2], 1.0,
                                                       normals[:, 0, c], uniforms[:, 0, c])
        assert np.array_equal(b[:, 0, c], ref_b), f"accept bits differ for chain {c}"
        np.testing.assert_allclose(s[:, 0, c], ref_s, rtol=1e-12, atol=0)
        tot_acc += acc
    assert summ[0]["acceptance"] == pytest.approx(tot_acc / (n_steps * n_chains), abs=1e-12)


@pytest.mark.parametrize("mode", ["recip_precise", "recip_fast", "nodiv_fast", "cross_fast", "log_fast"])
@pytest.mark.parametrize("target", [R2WS_REF, R2GAUSS_1, (O.WS, 2.0, 0.54), (O.GAUSS, 1.5, 0.0)])
def test_fp32_gpu_rule_injected_matches_a5(dev, oracle, mode, target):
    """a5/a7: Metropolis.cuh:40-52 with injected draws, n_coords coordinates per
    chain (the Gibbs sweep); accept bits identical, positions to 1e-6.  A chain is left out of the bit-for-bit
    check only if it comes closer to an accept/reject tie than the arithmetic of the mode can resolve: 2e-6 in
    PRECISE math (OCML expf vs glibc expf: last ulp), 1e-5 in FAST math (v_exp_f32 / v_rcp_f32 / v_log_f32);
    at ~3000 proposals per chain that is ~1 % / ~6 % of the chains."""
    from mcmc_amd import lib as L
    rule, math = {"recip_precise": (L.ACCEPT_RECIP, L.MATH_PRECISE), "recip_fast": (L.ACCEPT_RECIP, L.MATH_FAST),
                  "nodiv_fast": (L.ACCEPT_NODIV, L.MATH_FAST), "cross_fast": (L.ACCEPT_CROSS, L.MATH_FAST),
                  "log_fast": (L.ACCEPT_LOG, L.MATH_FAST)}[mode]
    tie = 2e-6 if math == L.MATH_PRECISE else 1e-5
    n_steps, n_chains, n_coords, step = 1024, 16, 3, 2.2
    rng = np.random.default_rng(99)
    normals = rng.standard_normal((n_steps, n_coords, n_chains)).astype(np.float32)
    uniforms = (1.0 - rng.random((n_steps, n_coords, n_chains))).astype(np.float32)  # (0,1]
    eng = _engine(dev, n_chains=n_chains, n_coords=n_coords, target=target, step=step, precision=32,
                  accept_rule=rule, math_mode=math)
    samples = torch.empty((n_steps, n_coords, n_chains), dtype=torch.float32, device=dev)
    bits = torch.zeros((n_steps, n_coords, n_chains), dtype=torch.uint8, device=dev)
    eng.run(n_steps, samples=samples, inj_normals=torch.from_numpy(normals).to(dev),
            inj_uniforms=torch.from_numpy(uniforms).to(dev), accept_bits=bits)
    torch.cuda.synchronize()
    s, b = samples.cpu().numpy(), bits.cpu().numpy()
    tg = [target] * n_coords
    strict = 0
    for c in range(n_chains):
        draws = np.stack([normals[:, :, c], uniforms[:, :, c]], axis=-1)  # [step][coord][2]
        x, ref_b, trace, margin = oracle.warmup_thread_trace(tg, [1.0] * n_coords, step, n_steps, draws)
        if margin > tie:
            assert np.array_equal(b[:, :, c], ref_b), f"accept bits differ for chain {c} (margin {margin})"
            np.testing.assert_allclose(s[:, :, c], trace, rtol=1e-6, atol=0)
            strict += 1
    # PRECISE: at most one chain may be left out; FAST: >= 85 %
    floor = n_chains - 1 if math == L.MATH_PRECISE else int(np.ceil(0.85 * n_chains))
    print(f"strict/total = {strict}/{n_chains}")
    assert strict >= floor, f"only {strict} of {n_chains} chains stayed clear of a tie and were checked bit for bit"
    final = eng.state.cpu().numpy()
    np.testing.assert_allclose(final, s[-1], rtol=0, atol=0)


def test_philox_draws_match_oracle(dev, oracle):
    """RNG layer: device Philox + packed fields + Box-Muller vs the oracle's independent restatement
    (all three streams; offsets chosen so that groups of 8 steps are cut at both ends)."""
    from mcmc_amd import lib as L
    n_chains, n_steps, coord, seed, off = 64, 45, 2, 314159265, 5
    for math, tol in ((L.MATH_PRECISE, 2e-6), (L.MATH_FAST, 2e-5)):
        bufs = [torch.zeros((n_steps, n_chains), dtype=torch.float32, device=dev) for _ in range(6)]
        L.check(L.lib.mcmc_launch_draws(seed, 1000, n_chains, coord, off, n_steps, math,
                                        *[C.c_void_p(t.data_ptr()) for t in bufs], None))
        torch.cuda.synchronize()
        z, u, a0, a1, sz, su = [t.cpu().numpy() for t in bufs]
        assert u.min() > 0.0 and u.max() < 1.0
        for c in (0, 5, 63):
            for i in range(n_steps):
                zn, un = oracle.mh_draws_f32(seed, 1000 + c, coord, off + i)
                b0, b1 = oracle.angle_draws_f32(seed, 1000 + c, coord, off + i)
                szn, sun = oracle.sab_mh_draws_f32(seed, 1000 + c, coord, off + i)
                # integer -> float paths are exact
                assert u[i, c] == un and a0[i, c] == b0 and a1[i, c] == b1 and su[i, c] == sun
                assert abs(z[i, c] - zn) <= tol * max(1.0, abs(zn))
                assert abs(sz[i, c] - szn) <= tol * max(1.0, abs(szn))


@pytest.mark.parametrize("t0", [0, 40, (2 ** 32 // 3) * 8 - 72])
def test_precomputed_philox_rounds_equal_the_generic_blocks(dev, t0):
    """Full groups of the FAST kernels draw through a PhiloxStream (rounds 1-2 worked out once per chain); one-step
    launches are partial groups and take the generic block.  Same trajectory bit for bit -- also where the block index
    3 g + j crosses 2^32 (the stream's counter word changes: t0 = 11453246048, crossing 72 steps later)."""
    from mcmc_amd import lib as L
    kw = dict(n_chains=300, n_coords=2, target=R2WS_REF, step=2.48, precision=32, accept_rule=L.ACCEPT_CROSS,
              math_mode=L.MATH_FAST, seed=99)
    a, b = _engine(dev, **kw), _engine(dev, **kw)
    a.steps_done = b.steps_done = t0
    n = 160
    a.run(n)
    for _ in range(n):
        b.run(1)
    torch.cuda.synchronize()
    assert a.steps_done == b.steps_done == t0 + n
    assert torch.equal(a.state, b.state)
    assert not torch.equal(a.state, torch.ones_like(a.state))


def test_samples_only_writeback_kernel_equals_the_general_one(dev):
    """FAST-math Philox launches that write samples but no accept flags take an instantiation of their own (no run-time
    tests in the step, row base + 32-bit lane offset): same samples, states and statistics as the general write-back
    kernel, for a ragged chain count over many tiles and a launch that starts mid-way through a draw group."""
    from mcmc_amd import lib as L
    N, n = 2048 * 256 + 300 + 77, 21
    kw = dict(n_chains=N, n_coords=2, target=R2WS_REF, step=2.48, precision=32, accept_rule=L.ACCEPT_CROSS,
              math_mode=L.MATH_FAST, seed=5, center=2.3)
    a, b = _engine(dev, **kw), _engine(dev, **kw)
    a.run(3), b.run(3)
    sa = torch.full((n, 2, N), -1.0, dtype=torch.float32, device=dev)
    sb = torch.full((n, 2, N), -2.0, dtype=torch.float32, device=dev)
    bits = torch.zeros((n, 2, N), dtype=torch.uint8, device=dev)
    ta = a.run(n, samples=sa).clone()
    tb = b.run(n, samples=sb, accept_bits=bits).clone()
    torch.cuda.synchronize()
    assert torch.equal(sa, sb) and torch.equal(a.state, b.state) and torch.equal(ta, tb)
    assert torch.equal(sa[-1], a.state)


def test_pair_layout_writeback_holds_the_same_samples(dev):
    """Option MCMC_SAMPLES_PAIRS (mcmc_hip.h): samples[n / 2][coords][chains][2], two consecutive steps of a chain as one
    64-bit store -- the default layout's values bit for bit (ragged chain count, many tiles, launch at an even offset inside
    a draw group); odd step counts / offsets and launches the layout does not cover are refused."""
    from mcmc_amd import lib as L
    N, n = 2048 * 256 + 300 + 77, 22
    kw = dict(n_chains=N, n_coords=2, target=R2WS_REF, step=2.48, precision=32, accept_rule=L.ACCEPT_CROSS,
              math_mode=L.MATH_FAST, seed=5, center=2.3)
    a, b = _engine(dev, **kw), _engine(dev, **kw)
    a.run(6), b.run(6)
    sa = torch.full((n, 2, N), -1.0, dtype=torch.float32, device=dev)
    sb = torch.full((n // 2, 2, N, 2), -2.0, dtype=torch.float32, device=dev)
    ta = a.run(n, samples=sa).clone()
    L.check(L.lib.mcmc_set_option(b"MCMC_SAMPLES_PAIRS", b"1"))
    try:
        tb = b.run(n, samples=sb).clone()
        torch.cuda.synchronize()
        with pytest.raises(L.McmcError):
            b.run(3, samples=sb)                      # odd step count
        with pytest.raises(L.McmcError):
            b.run(4, samples=sb, accept_bits=torch.zeros((4, 2, N), dtype=torch.uint8, device=dev))
    finally:
        L.lib.mcmc_set_option(b"MCMC_SAMPLES_PAIRS", None)
    # [n/2][c][N][2] -> [n/2][2][c][N] -> [n][c][N]
    assert torch.equal(sb.permute(0, 3, 1, 2).reshape(n, 2, N), sa)
    assert torch.equal(a.state, b.state) and torch.equal(ta, tb)
    c = _engine(dev, **kw)
    c.run(7)                                          # an odd offset under the default layout is fine
    c.run(2, samples=sa[:2])


def test_philox_mode_trajectory_matches_oracle_on_device_draws(dev, oracle):
    """Philox mode == injected mode fed with the library's own draws (logic parity at any seed)."""
    from mcmc_amd import lib as L
    n_chains, n_steps, seed = 64, 200, 42
    z = torch.zeros((n_steps, n_chains), dtype=torch.float32, device=dev)
    u = torch.zeros((n_steps, n_chains), dtype=torch.float32, device=dev)
    L.check(L.lib.mcmc_launch_draws(seed, 0, n_chains, 0, 0, n_steps, L.MATH_PRECISE, C.c_void_p(z.data_ptr()),
                                    C.c_void_p(u.data_ptr()), None, None, None, None, None))
    eng = _engine(dev, n_chains=n_chains, n_coords=1, target=R2WS_REF, step=2.2, precision=32,
                  accept_rule=L.ACCEPT_RECIP, math_mode=L.MATH_PRECISE, seed=seed)
    samples = torch.empty((n_steps, 1, n_chains), dtype=torch.float32, device=dev)
    bits = torch.zeros((n_steps, 1, n_chains), dtype=torch.uint8, device=dev)
    eng.run(n_steps, samples=samples, accept_bits=bits)
    torch.cuda.synchronize()
    zc, uc, s, b = z.cpu().numpy(), u.cpu().numpy(), samples.cpu().numpy(), bits.cpu().numpy()
    for c in range(0, n_chains, 9):
        draws = np.stack([zc[:, c], uc[:, c]], axis=-1).reshape(n_steps, 1, 2)
        x, ref_b, trace, margin = oracle.warmup_thread_trace([R2WS_REF], [1.0], 2.2, n_steps, draws)
        if margin > 1e-5:
            assert np.array_equal(b[:, 0, c], ref_b[:, 0])
            np.testing.assert_allclose(s[:, 0, c], trace[:, 0], rtol=1e-6)


@pytest.mark.parametrize("n_chains", [1, 3])
def test_streaming_statistics_match_a4(dev, oracle, n_chains):
    """a4: statistics.hpp:6-57 from the streaming accumulators; n_chains == 1 is
    the reference's case, n_chains > 1 the pooled generalisation."""
    from mcmc_amd import lib as L
    n_steps, n_lags = 5000, 49
    eng = _engine(dev, n_chains=n_chains, n_coords=1, target=R2WS_REF, step=2.5, precision=64,
                  accept_rule=L.ACCEPT_SHORTCIRCUIT, math_mode=L.MATH_PRECISE, n_lags=n_lags, center=2.3)
    samples = torch.empty((n_steps, 1, n_chains), dtype=torch.float64, device=dev)
    eng.run(n_steps, samples=samples)
    torch.cuda.synchronize()
    s = samples.cpu().numpy()[:, 0, :]
    summ = eng.summary()[0]
    if n_chains == 1:
        x = s[:, 0]
        assert summ["mean"] == pytest.approx(oracle.average_f64(x), rel=1e-6)
        assert summ["variance"] == pytest.approx(oracle.variance_f64(x), rel=1e-5)
        for k in range(1, n_lags + 1):
            assert summ["rho"][k - 1] == pytest.approx(oracle.autocorrelation_f64(x, k), abs=2e-5)
    else:
        N = s.size
        mu = s.mean()
        var = ((s - mu) ** 2).sum() / (N - 1)
        assert summ["mean"] == pytest.approx(mu, rel=1e-6)
        assert summ["variance"] == pytest.approx(var, rel=1e-5)
        for k in (1, 2, 10, 49):
            cov = ((s[:-k] - mu) * (s[k:] - mu)).sum() / (n_chains * (n_steps - k) - 1)
            assert summ["rho"][k - 1] == pytest.approx(cov / var, abs=2e-5)


def test_split_launches_and_chain_partition_are_invariant(dev):
    """Counter-based RNG: (a) one launch of n steps == two launches of n/2 (+ring carry);
    (b) chains [0,N) on one device == two half partitions with chain_offset (the multi-GPU sharding)."""
    from mcmc_amd import lib as L
    N, n, n_lags = 1000, 300, 16
    kw = dict(n_coords=2, target=R2WS_REF, step=2.2, precision=32, accept_rule=L.ACCEPT_RECIP, math_mode=L.MATH_FAST,
              n_lags=n_lags, center=2.3, seed=7)
    a = _engine(dev, n_chains=N, **kw)
    sa = a.run(n).clone()
    b = _engine(dev, n_chains=N, carry_ring=True, **kw)
    s1 = b.run(150, window_start=True).clone()
    s2 = b.run(150, window_start=False).clone()
    torch.cuda.synchronize()
    assert torch.equal(a.state, b.state)
    core = 3 + n_lags
    # the float kernels keep the lag products in FP32 per launch (mcmc_hip.h): 1000 chains x 300 products of |y y'| <= 6
    # give each sum a rounding scatter of ~1e-3 in absolute terms, whatever its size (high lags nearly cancel)
    np.testing.assert_allclose((s1 + s2)[:, :core].cpu().numpy(), sa[:, :core].cpu().numpy(), rtol=1e-5, atol=2e-3)
    np.testing.assert_allclose(s1[:, core:core + n_lags].cpu().numpy(), sa[:, core:core + n_lags].cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(s2[:, core + n_lags:].cpu().numpy(), sa[:, core + n_lags:].cpu().numpy(), rtol=1e-6)
    # partition
    h0 = _engine(dev, n_chains=N // 2, chain_offset=0, **kw)
    h1 = _engine(dev, n_chains=N // 2, chain_offset=N // 2, **kw)
    t0 = h0.run(n).clone()
    t1 = h1.run(n).clone()
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([h0.state, h1.state], dim=1), a.state)
    # sums, accepts, head and tail sums: per-chain FP64 -- equal to FP64 rounding.  The FP32 lag products are pooled per
    # wavefront of 64 chains in FP32 tiles (chain_lag_kernel.hpp): a partition at chain 500 regroups the chains, so they
    # agree to FP32 accumulation rounding (partitions at multiples of 64 chains: test_gpu_multi.py, FP64 rounding)
    tt, ss = (t0 + t1).cpu().numpy(), sa.cpu().numpy()
    np.testing.assert_allclose(tt[:, :3], ss[:, :3], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(tt[:, core:], ss[:, core:], rtol=0, atol=2e-6 * N)   # row sums of the (h, l) halves, FP32 tiles
    np.testing.assert_allclose(tt[:, 3:core], ss[:, 3:core], rtol=0, atol=2e-6 * N * n)


@pytest.mark.parametrize("mode", ["recip_fast", "nodiv_fast", "log_fast", "cross_fast", "recip_precise"])
def test_known_answer_r2ws_large(dev, mode):
    """Full-size property test: 2^20 chains x 3 coords; <r^2> of r^2 WS(2.0,0.54) = 6.424247."""
    from mcmc_amd import lib as L
    rule, math = {"recip_precise": (L.ACCEPT_RECIP, L.MATH_PRECISE), "recip_fast": (L.ACCEPT_RECIP, L.MATH_FAST),
                  "nodiv_fast": (L.ACCEPT_NODIV, L.MATH_FAST), "log_fast": (L.ACCEPT_LOG, L.MATH_FAST),
                  "cross_fast": (L.ACCEPT_CROSS, L.MATH_FAST)}[mode]
    N, n = 1 << 20, 400
    eng = _engine(dev, n_chains=N, n_coords=3, target=R2WS_REF, step=2.2, precision=32, accept_rule=rule,
                  math_mode=math, n_lags=0, center=0.0)
    eng.run(300)  # warm-up
    eng.run(n)
    torch.cuda.synchronize()
    for c, sm in enumerate(eng.summary()):
        ex2 = sm["variance"] + sm["mean"] ** 2
        # sd(x^2) ~ 5.3; tau_int(x^2) < 10 at step 2.2 -> 3 sigma < 3*5.3*sqrt(2*10/(N*n)) = 3.5e-3
        assert abs(ex2 - 6.424247) < 3.5e-3, (mode, c, ex2)
        assert 0.40 < sm["acceptance"] < 0.50


def test_argument_errors(dev):
    from mcmc_amd import api, lib as L
    eng = _engine(dev, n_chains=10, n_coords=1, target=R2WS_REF, step=1.0, n_lags=10)
    with pytest.raises(L.McmcError):
        eng.run(5)  # window start needs n_steps >= n_lags
    eng.args.workspace_bytes = 0
    with pytest.raises(L.McmcError):
        eng.run(50)


def test_tune_stepsize_reproduces_reference_optimum(dev):
    """mcmc_tune_stepsize on r^2 WS(2.0, 0.54): the reference's own study has its efficiency optimum at width 2.48
    (stepsize/results/r2WS_2.0_0.54.png, StepSizeAnalysis.ipynb); acceptance falls monotonically with the width."""
    from mcmc_amd import api, lib as L
    best, w, acc, tau = api.TuneStepSize(api.make_target(L.TARGET_R2WS, 2.0, 0.54), 1.0, 4.0, n_widths=13,
                                         n_chains=1 << 15, n_warmup=500, n_steps=1500)
    assert 2.2 < best < 2.8
    assert np.all(np.diff(acc) < 0) and np.all(tau >= 1.0)
    assert tau.min() < tau[0] and tau.min() < tau[-1]
    with pytest.raises(L.McmcError):
        api.TuneStepSize(api.make_target(L.TARGET_R2WS, 2.0, 0.54), 2.0, 1.0)


def test_bench_size_properties(dev):
    """BASELINE's full size (2^24 chains x 3 coordinates, the bench.py workload, CROSS rule / FAST math) through
    size-independent properties: one launch == two half launches (states bit-identical, statistics additive);
    slices of the id space re-run as small engines with chain_offset (first, an unaligned middle, the last 100
    chains) reproduce the big run's states bit for bit; accepted-proposal count == steps * chains * acceptance;
    <r^2> = 6.424247 within 3 standard errors."""
    from mcmc_amd import lib as L
    N, n = 1 << 24, 64
    kw = dict(n_coords=3, target=R2WS_REF, step=2.48, precision=32, accept_rule=L.ACCEPT_CROSS, math_mode=L.MATH_FAST,
              n_lags=0, center=2.3, seed=314159265)
    a = _engine(dev, n_chains=N, **kw)
    sa = a.run(n).clone()
    b = _engine(dev, n_chains=N, **kw)
    s1 = b.run(n // 2, window_start=True).clone()
    s2 = b.run(n // 2, window_start=False).clone()
    torch.cuda.synchronize()
    assert torch.equal(a.state, b.state)
    # the float partial sums are folded into FP64 at different steps in the two schedules: equal to rounding
    np.testing.assert_allclose((s1 + s2).cpu().numpy(), sa.cpu().numpy(), rtol=1e-6)
    assert (s1 + s2)[:, 2].cpu().tolist() == sa[:, 2].cpu().tolist()      # accepted proposals: exact
    for off, cnt in ((0, 300), (5_000_123, 777), (N - 100, 100)):
        h = _engine(dev, n_chains=cnt, chain_offset=off, **kw)
        h.run(n)
        torch.cuda.synchronize()
        assert torch.equal(h.state, a.state[:, off:off + cnt]), off
    st = sa.cpu().numpy()
    assert np.all(st[:, 2] == np.round(st[:, 2])) and np.all(st[:, 2] <= N * n)
    # equilibrium estimate after a warm-up (the first launch started every chain at x0 = 1)
    a.run(400)
    a.run(200, window_start=True)
    torch.cuda.synchronize()
    for sm in a.summary():
        ex2 = sm["variance"] + sm["mean"] ** 2
        assert abs(ex2 - 6.424247) < 3 * 5.3 * np.sqrt(2 * 6.0 / (N * 200.0)) + 2e-4, ex2
        assert 0.40 < sm["acceptance"] < 0.42


def test_randomised_configurations_against_oracle_stream(dev, oracle):
    """Differential test over 24 seeded random configurations (density, rule, width, 64-bit seed, chain offsets up
    to 2^39, step offsets up to 2^40 that start and end inside a group of 8 packed steps, 1..40 steps, ragged chain
    counts, a second launch continuing the first): the Philox-mode trajectory of sampled chains == the oracle's
    step logic driven by the ORACLE's own restatement of the packed draw stream (orc_mh_draws_f32)."""
    from mcmc_amd import lib as L
    rng = np.random.default_rng(20260101)
    kinds = [R2WS_REF, R2GAUSS_1, (O.WS, 2.0, 0.54), (O.GAUSS, 1.5, 0.0)]
    rules = [(L.ACCEPT_RECIP, L.MATH_PRECISE, 1e-6), (L.ACCEPT_RECIP, L.MATH_FAST, 2e-4), (L.ACCEPT_NODIV, L.MATH_FAST, 2e-4),
             (L.ACCEPT_CROSS, L.MATH_FAST, 2e-4), (L.ACCEPT_LOG, L.MATH_FAST, 2e-4)]
    strict = total = 0
    for case in range(24):
        tgt = kinds[case % 4]
        rule, math, tol = rules[case % 5]
        n_chains = int(rng.choice([1, 63, 65, 257, 300]))
        n_coords = int(rng.integers(1, 4))
        seed = int(rng.integers(0, 1 << 63))
        off = int(rng.integers(0, (1 << 39) - n_chains))
        t0 = int(rng.integers(0, 1 << 40))
        n1, n2 = int(rng.integers(1, 41)), int(rng.integers(1, 20))
        step = float(rng.uniform(0.5, 4.0))
        eng = _engine(dev, n_chains=n_chains, n_coords=n_coords, target=tgt, step=step, precision=32, accept_rule=rule,
                      math_mode=math, seed=seed, chain_offset=off)
        eng.steps_done = t0
        n = n1 + n2
        samples = torch.empty((n, n_coords, n_chains), dtype=torch.float32, device=dev)
        bits = torch.zeros((n, n_coords, n_chains), dtype=torch.uint8, device=dev)
        eng.run(n1, samples=samples[:n1], accept_bits=bits[:n1])
        eng.run(n2, samples=samples[n1:], accept_bits=bits[n1:])
        torch.cuda.synchronize()
        s, b = samples.cpu().numpy(), bits.cpu().numpy()
        for c in {0, n_chains - 1, n_chains // 2}:
            for k in range(n_coords):
                d = np.array([oracle.mh_draws_f32(seed, off + c, k, t0 + t) for t in range(n)], dtype=np.float32)
                _, rb, trace, margin = oracle.warmup_thread_trace([tgt], [1.0], step, n, d.reshape(n, 1, 2))
                total += 1
                if margin > 1e-4:
                    strict += 1
                    assert np.array_equal(b[:, k, c], rb[:, 0]), (case, c, k)
                    np.testing.assert_allclose(s[:, k, c], trace[:, 0], rtol=tol, atol=tol)
    # <= 60 proposals per chain and coordinate: a tie within 1e-4 is met by ~1 % of them
    print(f"strict/total = {strict}/{total}")
    assert strict >= 0.85 * total, (strict, total)


@pytest.mark.parametrize("n_steps", [1, 2, 5, 8])
def test_ragged_last_tile_with_writeback_over_many_tiles(dev, n_steps):
    """n_chains = 2^20 + 5: more tiles than blocks (the last tile runs on a late grid-stride iteration) and a last
    tile of which three waves lie wholly past the end.  Those waves must do nothing (they used to replay the last
    chain unsynchronised with its owner); the lanes beside the last chain replay it in lock-step.  The sample
    write-back and states of the last chains equal a small re-run of the same global ids."""
    from mcmc_amd import lib as L
    N = (1 << 20) + 5
    kw = dict(n_coords=2, target=R2WS_REF, step=2.48, precision=32, accept_rule=L.ACCEPT_CROSS, math_mode=L.MATH_FAST,
              n_lags=0, center=2.3, seed=77)
    for rep in range(3):     # a drifted wave shows only sometimes: a few independent launches
        a = _engine(dev, n_chains=N, **kw)
        a.steps_done = rep * 1000
        smp = torch.empty((n_steps, 2, N), dtype=torch.float32, device=dev)
        bits = torch.zeros((n_steps, 2, N), dtype=torch.uint8, device=dev)
        st = a.run(n_steps, samples=smp, accept_bits=bits).clone()
        h = _engine(dev, n_chains=70, chain_offset=N - 70, **kw)
        h.steps_done = rep * 1000
        smp_h = torch.empty((n_steps, 2, 70), dtype=torch.float32, device=dev)
        bits_h = torch.zeros((n_steps, 2, 70), dtype=torch.uint8, device=dev)
        h.run(n_steps, samples=smp_h, accept_bits=bits_h)
        torch.cuda.synchronize()
        assert torch.equal(smp[:, :, N - 70:], smp_h)
        assert torch.equal(bits[:, :, N - 70:], bits_h)
        assert torch.equal(a.state[:, N - 70:], h.state)
        assert torch.equal(a.state, smp[-1])
        # accepted proposals == the accept flags written back (every chain counted once, replay lanes not at all)
        assert st[:, 2].cpu().tolist() == bits.sum(dim=(0, 2), dtype=torch.float64).cpu().tolist()


def test_clock_probe_reports_a_plausible_shader_clock(dev):
    """mcmc_chain_clock_probe: the stamped twin of the bench kernel; MI355X runs 1.2 - 2.5 GHz under this load, and
    the stamped launch advances the chains exactly like mcmc_chain_run."""
    from mcmc_amd import api, lib as L
    kw = dict(n_coords=3, target=R2WS_REF, step=2.48, precision=32, accept_rule=L.ACCEPT_CROSS, math_mode=L.MATH_FAST,
              n_lags=0, center=2.3, seed=5)
    a = _engine(dev, n_chains=1 << 22, **kw)
    b = _engine(dev, n_chains=1 << 22, **kw)
    ghz = api.chain_clock_probe(a, 256)
    sb = b.run(256).clone()
    torch.cuda.synchronize()
    assert 1.2 < ghz < 2.6, ghz
    assert torch.equal(a.state, b.state) and torch.equal(a.stats, sb)
    with pytest.raises(L.McmcError):
        api.chain_clock_probe(_engine(dev, n_chains=64, n_coords=1, target=R2WS_REF, step=1.0, precision=64,
                                      accept_rule=L.ACCEPT_SHORTCIRCUIT, math_mode=L.MATH_PRECISE), 64)
