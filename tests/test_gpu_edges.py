"""GPU tests of the degenerate shapes of the entry points, through the C ABI: launches of zero steps, one chain,
one step, and launch lengths that are not a multiple of the 8-step draw group.  The counter-based RNG makes every one
of them checkable against an uninterrupted run (bit-exact state) and against the oracle on the device's own draws.
"""
import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu

R2WS_REF = (O.R2WS, 2.0, 0.54)      # PDF(), stepsize/include/metropolis.hpp:19-23


def _chain(dev, **kw):
    from mcmc_amd import api
    tk, p0, p1 = kw.pop("target")
    return api.ChainEngine(target=api.make_target(tk, p0, p1), device=dev, **kw)


@pytest.mark.parametrize("n_lags", [0, 16])
@pytest.mark.parametrize("precision", [32, 64])
def test_chain_zero_step_launch_is_a_no_op(dev, precision, n_lags):
    """n_steps = 0 (an empty batch): the state is untouched, the launch's statistics are all zero, and the chains
    continue exactly as if the empty launch had not happened (no n_lags: a window's first launch needs n_steps >= n_lags)."""
    from mcmc_amd import lib as L
    rule, math = (L.ACCEPT_SHORTCIRCUIT, L.MATH_PRECISE) if precision == 64 else (L.ACCEPT_RECIP, L.MATH_FAST)
    kw = dict(n_chains=777, n_coords=2, target=R2WS_REF, step=2.2, precision=precision, accept_rule=rule,
              math_mode=math, n_lags=n_lags, carry_ring=n_lags > 0, seed=11)
    a, b = _chain(dev, **kw), _chain(dev, **kw)
    a.run(40)
    b.run(40)
    before = a.state.clone()
    if n_lags == 0:
        st = a.run(0).clone()
        torch.cuda.synchronize()
        assert torch.equal(a.state, before)
        assert not st.any()
    else:
        with pytest.raises(L.McmcError):
            a.run(0)                      # window_start with fewer steps than lags is refused, nothing launched
        st = a.run(0, window_start=False).clone()
        torch.cuda.synchronize()
        assert torch.equal(a.state, before)
        assert not st[:, :3 + n_lags].any()
    sa = a.run(24, window_start=n_lags == 0).clone()
    sb = b.run(24, window_start=n_lags == 0).clone()
    torch.cuda.synchronize()
    assert torch.equal(a.state, b.state)
    assert torch.equal(sa, sb)


@pytest.mark.parametrize("n_steps", [1, 7, 9, 23])
def test_chain_one_chain_short_launches_follow_the_oracle_stream(dev, oracle, n_steps):
    """One chain, launch lengths below / across the 8-step draw group: the written samples equal a run of one-step
    launches bit for bit, and the oracle's trajectory (a5, Metropolis.cuh:40-52) on the library's own draws."""
    import ctypes as C
    from mcmc_amd import lib as L
    seed, step = 99, 2.5
    kw = dict(n_chains=1, n_coords=1, target=R2WS_REF, step=step, precision=32, accept_rule=L.ACCEPT_RECIP,
              math_mode=L.MATH_PRECISE, seed=seed)
    eng, one = _chain(dev, **kw), _chain(dev, **kw)
    smp = torch.empty((n_steps, 1, 1), dtype=torch.float32, device=dev)
    bits = torch.zeros((n_steps, 1, 1), dtype=torch.uint8, device=dev)
    eng.run(n_steps, samples=smp, accept_bits=bits)
    single = torch.empty((n_steps, 1, 1), dtype=torch.float32, device=dev)
    for t in range(n_steps):
        one.run(1, samples=single[t:t + 1])
    z = torch.zeros((n_steps, 1), dtype=torch.float32, device=dev)
    u = torch.zeros((n_steps, 1), dtype=torch.float32, device=dev)
    L.check(L.lib.mcmc_launch_draws(seed, 0, 1, 0, 0, n_steps, L.MATH_PRECISE, C.c_void_p(z.data_ptr()),
                                    C.c_void_p(u.data_ptr()), None, None, None, None, None))
    torch.cuda.synchronize()
    assert torch.equal(single, smp)
    assert torch.equal(one.state, eng.state)
    assert torch.equal(smp[-1], eng.state.reshape(1, 1))
    draws = np.stack([z.cpu().numpy()[:, 0], u.cpu().numpy()[:, 0]], axis=-1).reshape(n_steps, 1, 2)
    _, ref_b, trace, margin = oracle.warmup_thread_trace([R2WS_REF], [1.0], step, n_steps, draws)
    assert margin > 1e-5, "pick another seed: a tie within rounding of the accept test"
    assert np.array_equal(bits.cpu().numpy()[:, 0, 0], ref_b[:, 0])
    np.testing.assert_allclose(smp.cpu().numpy()[:, 0, 0], trace[:, 0], rtol=1e-6)


def test_xs_zero_iteration_warmup_and_refused_empty_integrals(dev):
    """WarmupMetropolis(0) leaves prevSample as it is; the integrals refuse samples = 0 (they divide by it:
    IntegrateR2.cuh:52, IntegrateSAB.cuh:88) without launching anything."""
    from mcmc_amd import api, lib as L
    nA = nB = 4
    eng = api.XsEngine(nA, nB, 300, [(L.TARGET_R2WS, 2.0, 0.54)] * (nA + nB), stepsize=2.2, device=dev)
    eng.WarmupMetropolis(16)
    before = eng.prevSample.clone()
    eng.WarmupMetropolis(0)
    torch.cuda.synchronize()
    assert torch.equal(eng.prevSample, before)
    with pytest.raises(L.McmcError):
        eng.MCIntegrate_R2(0)
    with pytest.raises(L.McmcError):
        eng.MCIntegrate_S_AB(0, 1.0)
    with pytest.raises(L.McmcError):
        eng.MCIntegrate_Inel(0, 1.0, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(eng.prevSample, before)
    # and the stream goes on where it stopped: equal to an engine that never saw the empty calls
    ref = api.XsEngine(nA, nB, 300, [(L.TARGET_R2WS, 2.0, 0.54)] * (nA + nB), stepsize=2.2, device=dev)
    ref.WarmupMetropolis(16)
    r_a, _ = eng.MCIntegrate_R2(8)
    r_b, _ = ref.MCIntegrate_R2(8)
    torch.cuda.synchronize()
    assert torch.equal(r_a, r_b)


def test_mvn_zero_sweep_launch_is_a_no_op(dev):
    """Stress kernel, n_steps = 0 inside an open window: state and ring untouched, statistics zero."""
    from mcmc_amd import api
    qd, qo = api.ar1_precision(64, 0.9)
    kw = dict(n_chains=600, q_diag=qd, q_off=qo, step=0.8, lag=5, seed=5, device=dev)
    a, b = api.MvnEngine(**kw), api.MvnEngine(**kw)
    a.run(12, window=(0, 1 << 20))
    b.run(12, window=(0, 1 << 20))
    s0, r0 = a.state.clone(), a.ring.clone()
    st = a.run(0, window=(0, 1 << 20)).clone()
    torch.cuda.synchronize()
    assert torch.equal(a.state, s0) and torch.equal(a.ring, r0)
    assert not st.any()
    sa = a.run(9, window=(0, 1 << 20)).clone()
    sb = b.run(9, window=(0, 1 << 20)).clone()
    torch.cuda.synchronize()
    assert torch.equal(a.state, b.state) and torch.equal(a.ring, b.ring)
    assert torch.equal(sa, sb)


def test_chain_writeback_rows_longer_than_2_gib(dev):
    """Samples-only write-back (buffer stores: the row length is the descriptor's 32-bit NUM_RECORDS, the lane offset a
    32-bit byte count): a row of 2^29 + 305 chains is 2 GiB + 1220 B, past the signed 32-bit range.  First, middle
    (around the 2 GiB byte offset) and last chains equal small re-runs of the same global ids; the last row is the state."""
    from mcmc_amd import lib as L
    N, n_steps = (1 << 29) + 305, 2
    kw = dict(n_coords=1, target=R2WS_REF, step=2.48, precision=32, accept_rule=L.ACCEPT_CROSS, math_mode=L.MATH_FAST,
              n_lags=0, center=2.3, seed=3)
    a = _chain(dev, n_chains=N, **kw)
    smp = torch.full((n_steps, 1, N), -5.0, dtype=torch.float32, device=dev)
    a.run(n_steps, samples=smp)
    torch.cuda.synchronize()
    assert torch.equal(a.state, smp[-1])
    for off in (0, (1 << 29) - 35, N - 70):
        h = _chain(dev, n_chains=70, chain_offset=off, **kw)
        smp_h = torch.empty((n_steps, 1, 70), dtype=torch.float32, device=dev)
        h.run(n_steps, samples=smp_h)
        torch.cuda.synchronize()
        assert torch.equal(smp[:, :, off:off + 70], smp_h), off
    del smp, a
    torch.cuda.empty_cache()
