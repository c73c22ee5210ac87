"""GPU parity tests of the xsection path (WarmupMetropolis, MCIntegrate_R2,
MCIntegrate_S_AB, black-disk hits, run estimator) against the CPU oracle,
through the C ABI.  Injected draw streams follow the reference's consumption
order (Metropolis.cuh:43,47; IntegrateSAB.cuh:50-52)."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu

WS20 = (O.R2WS, 2.0, 0.54)  # main.cu:14-19


def _modes():
    from mcmc_amd import lib as L
    return {"recip_precise": (L.ACCEPT_RECIP, L.MATH_PRECISE), "recip_fast": (L.ACCEPT_RECIP, L.MATH_FAST),
            "nodiv_fast": (L.ACCEPT_NODIV, L.MATH_FAST), "cross_fast": (L.ACCEPT_CROSS, L.MATH_FAST),
            "log_fast": (L.ACCEPT_LOG, L.MATH_FAST)}


def _tie(mode):
    """Closest approach to an accept/reject tie below which a chain is left out of the bit-for-bit check: what the
    arithmetic of the mode can resolve (OCML expf vs glibc expf: last ulp; hardware exp / rcp / log in FAST math)."""
    return 2e-6 if mode == "recip_precise" else 1e-5


@pytest.fixture
def option():
    """mcmc_set_option for the duration of a test (the diagnostic switches of the S_AB kernels)."""
    from mcmc_amd import lib as L
    touched = []

    def setopt(name, value):
        touched.append(name)
        L.check(L.lib.mcmc_set_option(name.encode(), value.encode() if value is not None else None))
    yield setopt
    for name in touched:
        L.lib.mcmc_set_option(name.encode(), None)


def _xs(dev, nA, nB, T, targets, mode="recip_precise", **kw):
    from mcmc_amd import api
    rule, math = _modes()[mode]
    return api.XsEngine(nA, nB, T, targets, accept_rule=rule, math_mode=math, device=dev, **kw)


def _draws(rng, T, L):
    """Per-thread streams of alternating (normal, uniform in (0,1])."""
    d = np.empty((T, L), dtype=np.float32)
    d[:, 0::2] = rng.standard_normal((T, L // 2)).astype(np.float32)
    d[:, 1::2] = (1.0 - rng.random((T, L // 2))).astype(np.float32)
    return d


@pytest.mark.parametrize("mode", ["recip_precise", "recip_fast", "nodiv_fast", "cross_fast", "log_fast"])
@pytest.mark.parametrize("nN", [1, 3, 8])
def test_warmup_and_r2_injected(dev, oracle, mode, nN):
    """a8 + a10: final state and per-thread Kahan means of r^2 (IntegrateR2.cuh:33-52)."""
    T, iters, samples, step = 70, 40, 64, 2.2
    targets = [WS20 if n % 2 == 0 else (O.R2WS, 2.54, 0.5) for n in range(nN)]
    rng = np.random.default_rng(5 + nN)
    d1 = _draws(rng, T, iters * nN * 2)
    d2 = _draws(rng, T, samples * nN * 2)
    xs = _xs(dev, nN, 0, T, targets, mode)
    xs.WarmupMetropolis(iters, stepsize=step, injected=torch.from_numpy(d1).to(dev))
    torch.cuda.synchronize()
    st1 = xs.prevSample.cpu().numpy().copy()
    res, run_mean = xs.MCIntegrate_R2(samples, injected=torch.from_numpy(d2).to(dev))
    torch.cuda.synchronize()
    st2, res, run_mean = xs.prevSample.cpu().numpy(), res.cpu().numpy(), run_mean.cpu().numpy()
    strict = 0
    for e in range(T):
        x1, _, _, m1 = oracle.warmup_thread_trace(targets, [1.0] * nN, step, iters, d1[e])
        _, _, _, m2 = oracle.warmup_thread_trace(targets, x1, step, samples, d2[e])
        x2, out = oracle.integrate_r2_thread(targets, x1, step, samples, d2[e])
        if min(m1, m2) <= _tie(mode):
            continue
        strict += 1
        np.testing.assert_allclose(st1[:, e], x1, rtol=1e-6)
        np.testing.assert_allclose(st2[:, e], x2, rtol=1e-6)
        np.testing.assert_allclose(res[:, e], out, rtol=2e-6)
    # (40 + 64) sweeps x nN proposals per event: a tie within the margin is met by < 2 % of the events
    print(f"strict/total = {strict}/{T}")
    assert strict >= (T - 1 if mode == "recip_precise" else T * 0.9), (strict, T)
    # device run mean (replaces host Average, main.cu:221-223) vs the oracle's FP32 Kahan average
    for n in range(nN):
        assert run_mean[n] == pytest.approx(oracle.average_f32(res[n]), rel=1e-6)


@pytest.mark.parametrize("mode", ["recip_precise", "recip_fast", "nodiv_fast", "cross_fast", "log_fast"])
@pytest.mark.parametrize("nA,nB", [(2, 2), (4, 4), (3, 5), (12, 12), (16, 20), (70, 70), (208, 208), (197, 63)])
@pytest.mark.parametrize("b", [0.0, 1.5, 5.0])   # 5.0: nuclei that only graze -- the pair culling of G = 64 is active
def test_sab_injected(dev, oracle, mode, nA, nB, b):
    """a12: per-thread Kahan mean of prod (1 - Gamma), plus black-disk hit counters."""
    nN = nA + nB
    T = 37 if nN < 60 else 9
    samples, step = (17 if nN < 60 else 6), 2.2
    targets = [WS20] * nN
    rng = np.random.default_rng(nA * 100 + nB)
    L = samples * nN * 4
    d = np.empty((T, L), dtype=np.float32)
    blk = d.reshape(T, samples, 2, nN, 2)
    blk[:, :, 0, :, 0] = rng.standard_normal((T, samples, nN))
    blk[:, :, 0, :, 1] = 1.0 - rng.random((T, samples, nN))
    blk[:, :, 1, :, :] = 1.0 - rng.random((T, samples, nN, 2))
    dmax2 = 4.3 / np.pi  # d < 2 r_N with pi (2 r_N)^2 /4... black disk: sigma_NN = pi d_max^2
    xs = _xs(dev, nA, nB, T, targets, mode)
    res, mean, hits = xs.MCIntegrate_S_AB(samples, b, stepsize=step, injected=torch.from_numpy(d).to(dev), dmax2=dmax2)
    torch.cuda.synchronize()
    res, mean, hits, st = res.cpu().numpy(), mean.cpu().numpy(), hits.cpu().numpy(), xs.prevSample.cpu().numpy()
    tot = np.zeros(3)
    strict = 0
    all_ok = True
    for e in range(T):
        mh = d[e].reshape(samples, 2, nN, 2)[:, 0].reshape(samples, nN * 2)
        _, _, _, margin = oracle.warmup_thread_trace(targets, [1.0] * nN, step, samples, mh)
        x, m, h, _ = oracle.integrate_sab_thread(nA, nB, targets, [1.0] * nN, step, samples, b, d[e], dmax2=dmax2)
        tot += h
        if margin <= _tie(mode):
            all_ok = False
            continue
        strict += 1
        np.testing.assert_allclose(st[:, e], x, rtol=1e-6)
        tol = 2e-5 if mode == "recip_precise" else 2e-4
        assert res[e] == pytest.approx(m, rel=tol, abs=1e-7), (e, res[e], m)
    # samples x nN proposals per event (<= 2500 for Pb-Pb): <= 5 % of the events meet a tie within the FAST margin
    print(f"strict/total = {strict}/{T}")
    assert strict >= (T - 1 if mode == "recip_precise" else np.floor(T * 0.85)), (strict, T)
    assert mean[0] == pytest.approx(res.astype(np.float64).mean(), rel=1e-12)
    assert hits[3] == T * samples
    if all_ok:
        # hit flags compare d^2 < dmax2 on positions equal to ~1e-7: allow a few borderline pairs
        assert abs(hits[0] - tot[0]) <= max(2, 1e-4 * tot[0])
        assert abs(hits[1] - tot[1]) <= 1
        assert abs(hits[2] - tot[2]) <= max(2, 1e-4 * tot[2])


@pytest.mark.parametrize("mode", ["recip_fast", "nodiv_fast", "cross_fast"])
@pytest.mark.parametrize("layout", ["species", "alternating", "all_distinct"])
def test_sab_per_nucleon_densities(dev, oracle, layout, mode):
    """pdfArr (main.cu:47-51) is one density per nucleon.  The S_AB kernel keeps the DISTINCT densities in LDS with a byte
    index per nucleon: 4 runs (protons / neutrons of A and B), 12 alternating runs and 14 distinct ones (more runs than
    table entries: the kernel reads the densities from global memory) all have to reproduce the oracle."""
    nA, nB = 6, 8
    nN = nA + nB
    if layout == "species":
        targets = [(O.R2WS, 2.0, 0.54)] * 3 + [(O.R2WS, 2.2, 0.5)] * 3 + [(O.R2WS, 2.54, 0.5)] * 4 + [(O.R2WS, 2.4, 0.6)] * 4
    elif layout == "alternating":
        targets = [(O.R2WS, 2.0, 0.54) if n % 2 == 0 or n >= 12 else (O.R2WS, 2.54, 0.5) for n in range(nN)]
    else:
        targets = [(O.R2WS, 2.0 + 0.05 * n, 0.5 + 0.01 * n) for n in range(nN)]
    T, samples, step, b = 41, 11, 2.2, 1.0
    rng = np.random.default_rng(5)
    d = np.empty((T, samples * nN * 4), dtype=np.float32)
    blk = d.reshape(T, samples, 2, nN, 2)
    blk[:, :, 0, :, 0] = rng.standard_normal((T, samples, nN))
    blk[:, :, 0, :, 1] = 1.0 - rng.random((T, samples, nN))
    blk[:, :, 1, :, :] = 1.0 - rng.random((T, samples, nN, 2))
    xs = _xs(dev, nA, nB, T, targets, mode)
    res, mean, _ = xs.MCIntegrate_S_AB(samples, b, stepsize=step, injected=torch.from_numpy(d).to(dev))
    torch.cuda.synchronize()
    res, st = res.cpu().numpy(), xs.prevSample.cpu().numpy()
    strict = 0
    for e in range(T):
        mh = d[e].reshape(samples, 2, nN, 2)[:, 0].reshape(samples, nN * 2)
        _, _, _, margin = oracle.warmup_thread_trace(targets, [1.0] * nN, step, samples, mh)
        x, m, _, _ = oracle.integrate_sab_thread(nA, nB, targets, [1.0] * nN, step, samples, b, d[e])
        if margin <= _tie(mode):
            continue
        strict += 1
        np.testing.assert_allclose(st[:, e], x, rtol=1e-6)
        assert res[e] == pytest.approx(m, rel=2e-4, abs=1e-7), (e, res[e], m)
    print(f"strict/total = {strict}/{T}")
    assert strict >= np.floor(T * 0.85), (strict, T)


@pytest.mark.parametrize("mode", ["recip_fast", "nodiv_fast", "cross_fast", "log_fast"])
@pytest.mark.parametrize("kinds", ["r2ws", "mixed"])
@pytest.mark.parametrize("nA,nB", [(6, 8), (70, 70)])
def test_sab_negative_start_and_mixed_kinds(dev, oracle, mode, kinds, nA, nB):
    """Metropolis.cuh:47-50,64 at its corners: (i) chains started at r < 0, where the r^2 densities are 0 and
    CopyLocalMHVars caches 1/0 = inf (every proposal onto p' > 0 is accepted, none onto p' = 0: 0 * inf >= U is false);
    the NODIV / CROSS step of the S_AB kernels carries e = +inf there (sab_aux_init); (ii) per-nucleon target kinds
    (uniform_kind = -1: the run-time switch inside sab_mh) under every rule.  Chain states, per-event S_AB and the
    inelastic flag against the oracle's reference rule."""
    nN = nA + nB
    T = 33 if nN < 60 else 9
    samples, step, b = (15 if nN < 60 else 6), 2.2, 1.5
    if kinds == "r2ws":
        targets = [WS20 if n % 3 else (O.R2WS, 2.54, 0.5) for n in range(nN)]
    else:
        cyc = [(O.R2WS, 2.0, 0.54), (O.WS, 2.0, 0.54), (O.R2GAUSS, 1.3, 0.0), (O.GAUSS, 1.1, 0.0)]
        targets = [cyc[n % 4] for n in range(nN)]
    # every other nucleon starts at r < 0 (and nucleon 0 always does); the rest at the reference's 1.0 (main.cu:132)
    x0 = np.array([(-0.4 - 0.3 * (n % 5)) if n % 2 == 0 else 1.0 for n in range(nN)], dtype=np.float32)
    rng = np.random.default_rng(nA * 7 + nB + (1 if kinds == "mixed" else 0))
    d = np.empty((T, samples * nN * 4), dtype=np.float32)
    blk = d.reshape(T, samples, 2, nN, 2)
    blk[:, :, 0, :, 0] = rng.standard_normal((T, samples, nN))
    blk[:, :, 0, :, 1] = 1.0 - rng.random((T, samples, nN))
    blk[:, :, 1, :, :] = 1.0 - rng.random((T, samples, nN, 2))
    dmax2 = 4.3 / np.pi
    dj = torch.from_numpy(d).to(dev)
    start = torch.from_numpy(np.repeat(x0[:, None], T, axis=1).copy()).to(dev)
    xs = _xs(dev, nA, nB, T, targets, mode)
    xs.prevSample.copy_(start)
    res, _, hits = xs.MCIntegrate_S_AB(samples, b, stepsize=step, injected=dj, dmax2=dmax2)
    flag = _xs(dev, nA, nB, T, targets, mode)
    flag.prevSample.copy_(start)
    inel = flag.MCIntegrate_Inel(samples, b, dmax2, stepsize=step, injected=dj).clone()
    torch.cuda.synchronize()
    res, hits, inel, st = res.cpu().numpy(), hits.cpu().numpy(), inel.cpu().numpy(), xs.prevSample.cpu().numpy()
    assert torch.equal(xs.prevSample, flag.prevSample)
    assert inel[0] == hits[1] and inel[1] == T * samples
    strict, tot, all_ok = 0, 0, True
    for e in range(T):
        mh = d[e].reshape(samples, 2, nN, 2)[:, 0].reshape(samples, nN * 2)
        _, bits, _, margin = oracle.warmup_thread_trace(targets, x0, step, samples, mh)
        x, m, h, _ = oracle.integrate_sab_thread(nA, nB, targets, x0, step, samples, b, d[e], dmax2=dmax2)
        tot += h[1]
        if margin <= _tie(mode):
            all_ok = False
            continue
        strict += 1
        np.testing.assert_allclose(st[:, e], x, rtol=1e-6)
        assert res[e] == pytest.approx(m, rel=2e-4, abs=1e-7), (e, res[e], m)
    print(f"strict/total = {strict}/{T}")
    assert strict >= np.floor(T * 0.85), (strict, T)
    if all_ok:
        assert abs(int(inel[0]) - int(tot)) <= 1


def test_sab_philox_matches_oracle_on_device_draws(dev, oracle):
    """Philox mode == injected mode fed with the library's own draws (He-4 + He-4, main.cu:38-41)."""
    from mcmc_amd import lib as L
    nA = nB = 4
    nN, T, samples, seed, b = 8, 16, 10, 99, 0.7
    per = []
    for n in range(nN):
        bufs = [torch.zeros((samples, T), dtype=torch.float32, device=dev) for _ in range(4)]
        a0b, a1b, zb, ub = [C.c_void_p(t.data_ptr()) for t in bufs]
        L.check(L.lib.mcmc_launch_draws(seed, 0, T, n, 0, samples, L.MATH_PRECISE, None, None, a0b, a1b, zb, ub, None))
        bufs = [bufs[2], bufs[3], bufs[0], bufs[1]]  # -> z, u, a0, a1 of the S_AB kernel's streams
        per.append([t.cpu().numpy() for t in bufs])
    torch.cuda.synchronize()
    d = np.empty((T, samples, 2, nN, 2), dtype=np.float32)
    for n in range(nN):
        z, u, a0, a1 = per[n]
        d[:, :, 0, n, 0] = z.T
        d[:, :, 0, n, 1] = u.T
        d[:, :, 1, n, 0] = a0.T
        d[:, :, 1, n, 1] = a1.T
    d = d.reshape(T, -1)
    xs = _xs(dev, nA, nB, T, [WS20] * nN, "recip_precise", seed=seed)
    res, mean, _ = xs.MCIntegrate_S_AB(samples, b, stepsize=2.2)
    torch.cuda.synchronize()
    res = res.cpu().numpy()
    for e in range(T):
        x, m, _, _ = oracle.integrate_sab_thread(nA, nB, [WS20] * nN, [1.0] * nN, 2.2, samples, b, d[e])
        assert res[e] == pytest.approx(m, rel=2e-5, abs=1e-7)


def test_gamma_nn_negative_at_zero(oracle):
    """F8: Gamma_NN(0) = 4.3/(4 pi 0.2) = 1.7109 > 1 is kept (no clamp)."""
    assert oracle.gamma_nn_r2(0.0) == pytest.approx(1.710916, rel=1e-6)


def test_layer_b_rms_known_answer(dev):
    """Layer B whole RMS integration (main.cu:157-286) vs quadrature: RMS of r^2 WS(2.0, 0.54) = 2.534610."""
    from mcmc_amd import lib as L
    cfg = L.Config()
    L.lib.mcmc_config_default(C.byref(cfg))
    cfg.samplesForMix = 512
    cfg.totalRuns = 8
    h = C.c_void_p()
    L.check(L.lib.mcmc_create(C.byref(cfg), C.byref(h)))
    try:
        L.check(L.lib.mcmc_warmup(h, 2.2, cfg.samplesForMix))
        rms = (C.c_float * 8)()
        err = (C.c_float * 8)()
        L.check(L.lib.mcmc_run_rms(h, rms, err))
        for n in range(8):
            assert err[n] > 0
            assert abs(rms[n] - 2.534610) < max(4 * err[n], 1e-3), (n, rms[n], err[n])
    finally:
        L.check(L.lib.mcmc_destroy(h))


def test_layer_b_sab_scan_and_cross_sections(dev):
    """Layer B S_AB scan (main.cu:289-413) on He-4 + He-4 at reduced size: S_AB rises from ~0 towards 1 with b,
    stderr > 0, and the C++ sigma integration runs on the table (f1)."""
    from mcmc_amd import api, lib as L
    cfg = L.Config()
    L.lib.mcmc_config_default(C.byref(cfg))
    cfg.totalBlocks, cfg.threadsPerBlock, cfg.samplesPerThread, cfg.totalRuns = 16, 128, 64, 4
    cfg.ndbs, cfg.db = 40, 0.25
    h = C.c_void_p()
    L.check(L.lib.mcmc_create(C.byref(cfg), C.byref(h)))
    try:
        L.check(L.lib.mcmc_warmup(h, 2.2, 256))
        n = cfg.ndbs
        b, S, E = (C.c_float * n)(), (C.c_float * n)(), (C.c_float * n)()
        pin = (C.c_double * n)()
        L.check(L.lib.mcmc_run_sab_scan(h, b, S, E, pin, 4.3 / np.pi))
        S = np.array(S[:])
        assert S[0] < 0.6 and S[-1] > 0.99
        assert np.all(np.array(E[:]) >= 0)
        assert pin[0] > 0.8 and pin[n - 1] < 0.05
        cs = api.CrossSections(np.array(b[:], dtype=np.float64), S.astype(np.float64), np.array(E[:], dtype=np.float64))
        assert 200 < cs["rxn"][0] < 2000
    finally:
        L.check(L.lib.mcmc_destroy(h))


@pytest.mark.parametrize("mode", ["recip_precise", "recip_fast", "nodiv_fast", "cross_fast", "log_fast"])
@pytest.mark.parametrize("nA,nB", [(2, 2), (3, 5), (16, 20), (70, 70), (208, 208)])
@pytest.mark.parametrize("b", [0.0, 3.0, 5.0])
def test_inel_injected(dev, oracle, mode, nA, nB, b):
    """Inelastic-flag-only kernel (mcmc_launch_integrate_inel): the count of samples with >= 1 black-disk
    collision equals the oracle's, the chains advance exactly as in MCIntegrate_S_AB, and the count is identical
    to hits[1] of the full-counter kernel on the same draws."""
    nN = nA + nB
    T = 37 if nN < 60 else 9
    samples, step = (17 if nN < 60 else 6), 2.2
    targets = [WS20] * nN
    rng = np.random.default_rng(nA * 100 + nB + 7)
    d = np.empty((T, samples * nN * 4), dtype=np.float32)
    blk = d.reshape(T, samples, 2, nN, 2)
    blk[:, :, 0, :, 0] = rng.standard_normal((T, samples, nN))
    blk[:, :, 0, :, 1] = 1.0 - rng.random((T, samples, nN))
    blk[:, :, 1, :, :] = 1.0 - rng.random((T, samples, nN, 2))
    dmax2 = 4.3 / np.pi
    dj = torch.from_numpy(d).to(dev)
    xs = _xs(dev, nA, nB, T, targets, mode)
    inel = xs.MCIntegrate_Inel(samples, b, dmax2, stepsize=step, injected=dj).clone()
    full = _xs(dev, nA, nB, T, targets, mode)
    _, _, hits = full.MCIntegrate_S_AB(samples, b, stepsize=step, injected=dj, dmax2=dmax2)
    torch.cuda.synchronize()
    inel, hits = inel.cpu().numpy(), hits.cpu().numpy()
    assert inel[1] == T * samples
    assert inel[0] == hits[1]
    assert torch.equal(xs.prevSample, full.prevSample)
    tot, all_ok = 0, True
    for e in range(T):
        mh = d[e].reshape(samples, 2, nN, 2)[:, 0].reshape(samples, nN * 2)
        _, _, _, margin = oracle.warmup_thread_trace(targets, [1.0] * nN, step, samples, mh)
        _, _, h, _ = oracle.integrate_sab_thread(nA, nB, targets, [1.0] * nN, step, samples, b, d[e], dmax2=dmax2)
        tot += h[1]
        all_ok = all_ok and margin > _tie(mode)
    if all_ok:
        assert abs(inel[0] - tot) <= 1
    else:
        assert abs(inel[0] - tot) <= 0.2 * T * samples


@pytest.mark.parametrize("b", [0.0, 13.0, 17.0])
def test_inel_heavy_ion_equals_full_counters(dev, b):
    """Pb-208 + Pb-208 (one event per wavefront, early exit at the first collision): Philox mode, central,
    peripheral and beyond-contact impact parameters; the flag count equals hits[1] of the full-counter kernel
    and the chain state is bit-identical."""
    from mcmc_amd import lib as L
    nA = nB = 208
    PB = (O.R2WS, 6.62, 0.546)
    T, samples = 24, 8
    kw = dict(seed=4242, event_offset=1000)
    xs = _xs(dev, nA, nB, T, [PB] * (nA + nB), "nodiv_fast", **kw)
    full = _xs(dev, nA, nB, T, [PB] * (nA + nB), "nodiv_fast", **kw)
    for eng in (xs, full):
        eng.WarmupMetropolis(300, stepsize=2.2)
    dmax2 = 6.76 / np.pi
    inel = xs.MCIntegrate_Inel(samples, b, dmax2).clone()
    _, _, hits = full.MCIntegrate_S_AB(samples, b, dmax2=dmax2)
    torch.cuda.synchronize()
    inel, hits = inel.cpu().numpy(), hits.cpu().numpy()
    assert inel[1] == T * samples and inel[0] == hits[1]
    assert torch.equal(xs.prevSample, full.prevSample)
    if b == 0.0:
        assert inel[0] == T * samples
    if b == 17.0:
        assert inel[0] < 0.5 * T * samples


@pytest.mark.parametrize("b", [11.0, 13.5, 15.0, 16.5, 18.0, 19.5])
def test_inel_peripheral_equals_full_counters(dev, b):
    """Peripheral Pb-208 + Pb-208 events in the inelastic-flag kernel (probe misses, eager angles, bounding-box culling, pair
    scan with early exit) over 16 step pairs of 512 events and two launches: the flag count equals hits[1] of the
    full-counter kernel exactly and the chains advance identically -- from grazing (p_inel ~ 1) to b = 19.5 fm (p_inel < 1 %)."""
    nA = nB = 208
    T, samples = 512, 32
    kw = dict(seed=20261112, event_offset=7)
    xs = _xs(dev, nA, nB, T, [PB] * (nA + nB), "cross_fast", **kw)
    full = _xs(dev, nA, nB, T, [PB] * (nA + nB), "cross_fast", **kw)
    for eng in (xs, full):
        eng.InitSampleFromPdf()
        eng.WarmupMetropolis(64, stepsize=3.9)
    dmax2 = 6.76 / np.pi
    total = 0
    for launch in range(2):     # the second launch starts with odd-numbered history: every event probes again first
        inel = xs.MCIntegrate_Inel(samples - launch, b, dmax2, stepsize=3.9).clone()
        _, _, hits = full.MCIntegrate_S_AB(samples - launch, b, stepsize=3.9, dmax2=dmax2)
        torch.cuda.synchronize()
        inel, hits = inel.cpu().numpy(), hits.cpu().numpy()
        assert inel[1] == T * (samples - launch) and inel[0] == hits[1], (b, launch, inel, hits)
        assert torch.equal(xs.prevSample, full.prevSample)
        total += inel[0]
    if b <= 13.5:
        assert total > 0.5 * T * (2 * samples - 1)
    if 15.0 <= b <= 18.0:
        assert 0 < total < 0.9 * T * (2 * samples - 1)      # collisions happen, and not always


def test_layer_b_inel_scan(dev):
    """mcmc_run_inel_scan == the p_inel column of mcmc_run_sab_scan on a handle with the same seed."""
    from mcmc_amd import lib as L
    out = []
    for which in ("inel", "sab"):
        cfg = L.Config()
        L.lib.mcmc_config_default(C.byref(cfg))
        cfg.totalBlocks, cfg.threadsPerBlock, cfg.samplesPerThread, cfg.totalRuns = 8, 128, 32, 3
        cfg.ndbs, cfg.db = 12, 0.6
        h = C.c_void_p()
        L.check(L.lib.mcmc_create(C.byref(cfg), C.byref(h)))
        try:
            L.check(L.lib.mcmc_warmup(h, 2.2, 128))
            n = cfg.ndbs
            b, pin, perr = (C.c_float * n)(), (C.c_double * n)(), (C.c_double * n)()
            if which == "inel":
                L.check(L.lib.mcmc_run_inel_scan(h, b, pin, perr, 4.3 / np.pi))
                assert all(e >= 0 for e in perr[:]) and max(perr[:]) > 0
            else:
                S, E = (C.c_float * n)(), (C.c_float * n)()
                L.check(L.lib.mcmc_run_sab_scan(h, b, S, E, pin, 4.3 / np.pi))
            out.append(np.array(pin[:]))
        finally:
            L.check(L.lib.mcmc_destroy(h))
    np.testing.assert_allclose(out[0], out[1], rtol=0, atol=1e-12)
    assert out[0][0] > 0.8 and out[0][-1] < 0.1


def test_init_from_pdf_matches_oracle_and_is_stationary(dev, oracle):
    """Randomised starts (main.cu:144-148's suggestion): the device's inverse-CDF draw equals the oracle's for the same
    Philox word (to table rounding), the starts already have the target's <r^2> (quadrature 6.424247 for
    r^2 WS(2.0, 0.54)), a short warm-up leaves it there, and the draw depends on the global event id only."""
    T = 1 << 16
    targets = [WS20, (O.R2WS, 6.62, 0.546), (O.R2GAUSS, 1.0, 0.0), (O.WS, 2.0, 0.54)]
    xs = _xs(dev, 4, 0, T, targets, "recip_fast", seed=123, event_offset=500)
    xs.InitSampleFromPdf()
    torch.cuda.synchronize()
    x = xs.prevSample.cpu().numpy().astype(np.float64)
    for n, (k, p0, p1) in enumerate(targets):
        for e in (0, 1, 255, 256, 40000, T - 1):
            assert x[n, e] == pytest.approx(oracle.init_from_pdf(k, p0, p1, 123, 500 + e, n), rel=2e-4, abs=1e-5)
    se = lambda m2, m4: np.sqrt((m4 - m2 * m2) / T)  # noqa: E731
    r2 = (x ** 2).mean(axis=1)
    assert r2[0] == pytest.approx(6.424247, abs=5 * se(r2[0], (x[0] ** 4).mean()) + 2e-3)
    assert r2[1] == pytest.approx(30.413841, abs=5 * se(r2[1], (x[1] ** 4).mean()) + 1e-2)
    assert r2[2] == pytest.approx(3.0, abs=5 * se(r2[2], (x[2] ** 4).mean()) + 2e-3)
    xs.WarmupMetropolis(16, stepsize=2.2)
    torch.cuda.synchronize()
    y = xs.prevSample.cpu().numpy().astype(np.float64)
    assert (y[0] ** 2).mean() == pytest.approx(6.424247, abs=5 * se(r2[0], (x[0] ** 4).mean()) + 2e-3)
    # partition: a second engine holding events [700, 700+64) draws the same starts
    part = _xs(dev, 4, 0, 64, targets, "recip_fast", seed=123, event_offset=700)
    part.InitSampleFromPdf()
    torch.cuda.synchronize()
    assert np.array_equal(part.prevSample.cpu().numpy(), x[:, 200:264].astype(np.float32))


def test_sab_heavy_ion_full_size_partition(dev):
    """C4 at the size of tools/run_c4.sh (Pb-208 + Pb-208, 2^14 events per launch): slices of the event id space
    re-run as small engines with event_offset reproduce the big launch's per-event S_AB, hit counters and chain
    states bit for bit (what makes the multi-GPU event partition exact), and S_AB rises with b."""
    nA = nB = 208
    PB = (O.R2WS, 6.62, 0.546)
    T, samples, dmax2 = 1 << 14, 4, 6.76 / np.pi
    big = _xs(dev, nA, nB, T, [PB] * 416, "nodiv_fast", seed=99, sigmaNN=6.76)
    big.InitSampleFromPdf()
    big.WarmupMetropolis(32, stepsize=3.9)
    start = big.prevSample.clone()
    res, mean, hits = big.MCIntegrate_S_AB(samples, 13.0, stepsize=3.9, dmax2=dmax2)
    torch.cuda.synchronize()
    res_big, hits_big, state_big = res.clone(), hits.clone(), big.prevSample.clone()
    tot = torch.zeros(3, dtype=torch.float64, device=dev)
    for off, cnt in ((0, 64), (7001, 100), (T - 50, 50)):
        part = _xs(dev, nA, nB, cnt, [PB] * 416, "nodiv_fast", seed=99, sigmaNN=6.76, event_offset=off)
        part.prevSample.copy_(start[:, off:off + cnt])
        part.args.step_offset = 32
        r, _, h = part.MCIntegrate_S_AB(samples, 13.0, stepsize=3.9, dmax2=dmax2)
        torch.cuda.synchronize()
        assert torch.equal(r, res_big[off:off + cnt]) and torch.equal(part.prevSample, state_big[:, off:off + cnt])
        tot += h[:3]
    assert torch.all(tot <= hits_big[:3]) and hits_big[3].item() == T * samples
    s = []
    for b in (11.0, 13.0, 15.0, 17.0):
        _, m, _ = big.MCIntegrate_S_AB(samples, b, stepsize=3.9)
        torch.cuda.synchronize()
        s.append(m.item())
    assert s[0] < s[1] < s[2] < s[3] and s[0] < 0.05 and s[3] > 0.9


def test_glauber_scan_event_sharding_is_exact(dev):
    """mcmc_amd.dist.glauber_scan: two engines holding the two halves of the event id space (what two ranks run)
    give, summed, the statistics tensor of one engine over all events -- inelastic counts exactly, S_AB sums to
    FP64 rounding -- for the full and the inelastic-only scan; finalize_scan then yields S_AB(b) rising with b."""
    from mcmc_amd import dist as D
    nA = nB = 12
    tg = [WS20] * (nA + nB)
    T, samples, runs, dmax2 = 600, 8, 2, 4.3 / np.pi
    b = np.array([0.0, 2.0, 4.0, 8.0])

    def scan(off, cnt, inel_only):
        e = _xs(dev, nA, nB, cnt, tg, "nodiv_fast", seed=11, event_offset=off)
        e.InitSampleFromPdf()
        e.WarmupMetropolis(64, stepsize=2.4)
        return D.glauber_scan(e, b, samples, runs, dmax2=dmax2, inel_only=inel_only)

    for inel_only in (False, True):
        whole = scan(0, T, inel_only)
        parts = scan(0, 250, inel_only) + scan(250, 350, inel_only)
        torch.cuda.synchronize()
        w, p = whole.cpu().numpy(), parts.cpu().numpy()
        assert np.array_equal(w[:, :, 1:], p[:, :, 1:])
        np.testing.assert_allclose(p[:, :, 0], w[:, :, 0], rtol=1e-12)
        res = D.finalize_scan(w, b)
        assert res["p_inel"][0] > 0.9 and res["p_inel"][-1] < 0.05 and "sigma_inel_mb" in res
        if not inel_only:
            assert res["S_AB"][0] < res["S_AB"][1] < res["S_AB"][2] < res["S_AB"][3] and res["S_AB"][3] > 0.98
            assert res["sigma"]["rxn"][0] > 0


@pytest.mark.parametrize("b", [0.0, 9.0, 13.0])
def test_sab_zero_product_exit_changes_nothing(dev, b):
    """Product-only S_AB for Pb-Pb stops an event's pair scan once its product has underflowed to 0; the kernel
    with hit counters scans every pair.  Same events, same draws: per-event results agree to below the smallest
    normal float (where the full scan ends on a denormal the early exit returns 0), chain states bit for bit."""
    nA = nB = 208
    PB = (O.R2WS, 6.62, 0.546)
    T, samples = 40, 4
    kw = dict(seed=777, sigmaNN=6.76)
    e1 = _xs(dev, nA, nB, T, [PB] * 416, "nodiv_fast", **kw)
    e3 = _xs(dev, nA, nB, T, [PB] * 416, "nodiv_fast", **kw)
    for e in (e1, e3):
        e.InitSampleFromPdf()
        e.WarmupMetropolis(16, stepsize=3.9)
    r1, _, _ = e1.MCIntegrate_S_AB(samples, b, stepsize=3.9)
    r3, _, hits = e3.MCIntegrate_S_AB(samples, b, stepsize=3.9, dmax2=6.76 / np.pi)
    torch.cuda.synchronize()
    a, c = r1.cpu().numpy(), r3.cpu().numpy()
    np.testing.assert_allclose(a, c, rtol=0, atol=1.2e-38)
    assert torch.equal(e1.prevSample, e3.prevSample)
    if b == 0.0:
        assert np.all(a == 0.0)
    if b == 13.0:
        assert a.max() > 0.01 and np.array_equal(a, c)


@pytest.mark.parametrize("mode", ["recip_fast", "nodiv_fast"])
@pytest.mark.parametrize("nA,nB", [(4, 4), (1, 1), (2, 3), (4, 1), (3, 4)])
def test_sab_light_nuclei_register_kernel_equals_general_kernel(dev, mode, nA, nB, option):
    """Light nuclei (<= 4 + 4 nucleons, the reference's default He-4 + He-4) take a kernel with one event per lane and
    the nucleons in registers.  Same draws, same step arithmetic and the same ORDER of the pair product as the general
    LDS kernel: chain states and per-event S_AB means bit-identical, for a ragged event count, an odd number of
    samples and a launch that starts on an odd sweep (so the kernel choice, which depends on a device's share of the
    events, cannot show in the results)."""
    nN = nA + nB
    T, samples, b = 1000, 9, 1.3
    tg = [WS20 if n % 2 == 0 else (O.R2WS, 1.7, 0.5) for n in range(nN)]
    out = []
    for small in ("1", "0"):
        option("MCMC_SAB_SMALL", small)
        e = _xs(dev, nA, nB, T, tg, mode, seed=31, event_offset=12345)
        e.WarmupMetropolis(5, stepsize=2.2)
        r1, m1, _ = e.MCIntegrate_S_AB(samples, b)
        r1 = r1.clone()
        r2, m2, _ = e.MCIntegrate_S_AB(4, b + 0.5)
        torch.cuda.synchronize()
        out.append((r1.cpu().numpy(), r2.cpu().numpy().copy(), e.prevSample.cpu().numpy().copy(), m2.item()))
    (a1, a2, sa, ma), (g1, g2, sg, mg) = out
    assert np.array_equal(sa, sg)
    assert np.array_equal(a1, g1) and np.array_equal(a2, g2)
    assert ma == mg
    assert 0.0 < a1.mean() < 1.0


def test_sab_light_nuclei_default_dispatch_is_result_neutral(dev, option):
    """Without the override the register kernel is taken from 2^17 events on (below that the general kernel is
    faster); whichever runs, chain states and per-event integrals are bit-identical."""
    T = 1 << 17
    out = []
    for small in (None, "0"):
        option("MCMC_SAB_SMALL", small)
        e = _xs(dev, 4, 4, T, [WS20] * 8, "recip_fast", seed=5)
        e.WarmupMetropolis(4, stepsize=2.2)
        r, m, _ = e.MCIntegrate_S_AB(8, 2.0)
        torch.cuda.synchronize()
        out.append((r.cpu().numpy().copy(), e.prevSample.cpu().numpy().copy(), m.item()))
    (a, sa, ma), (g, sg, mg) = out
    assert np.array_equal(sa, sg) and np.array_equal(a, g) and ma == mg


def test_sab_light_nuclei_rank_count_invariance_across_the_kernel_threshold(dev):
    """He-4 + He-4 (main.cu:38-41), 2^17 events: one device owning all of them takes the register kernel, two ranks
    owning 2^16 each (event_offset = first global id) take the general kernel.  Per-event S_AB means, chain states and
    the summed run statistic are bit-identical: results do not depend on the number of ranks."""
    T, samples, b = 1 << 17, 6, 1.1
    one = _xs(dev, 4, 4, T, [WS20] * 8, "recip_fast", seed=2025)
    one.WarmupMetropolis(8, stepsize=2.2)
    r, m, _ = one.MCIntegrate_S_AB(samples, b)
    torch.cuda.synchronize()
    halves = []
    for off in (0, T // 2):
        h = _xs(dev, 4, 4, T // 2, [WS20] * 8, "recip_fast", seed=2025, event_offset=off)
        h.WarmupMetropolis(8, stepsize=2.2)
        rh, mh, _ = h.MCIntegrate_S_AB(samples, b)
        torch.cuda.synchronize()
        halves.append((rh.clone(), h.prevSample.clone(), mh.item()))
    assert torch.equal(torch.cat([halves[0][0], halves[1][0]]), r)
    assert torch.equal(torch.cat([halves[0][1], halves[1][1]], dim=1), one.prevSample)
    assert 0.5 * (halves[0][2] + halves[1][2]) == pytest.approx(m.item(), rel=1e-14)


PB = (O.R2WS, 6.62, 0.546)


@pytest.mark.parametrize("nA,nB,b", [(208, 208, 0.0), (208, 208, 9.0), (208, 208, 13.0), (208, 208, 17.0), (208, 208, 40.0),
                                     (197, 63, 7.0), (63, 197, 11.0), (129, 40, 9.5)])
@pytest.mark.parametrize("mode", ["recip_fast", "recip_precise"])
def test_sab_exact_pair_culling_changes_nothing(dev, mode, nA, nB, b, option):
    """One event per wavefront: nucleons outside the other nucleus's bounding box grown by the distance at which
    1 - Gamma rounds to exactly 1.0f (and beyond the black-disk radius) are dropped before the pair tile.  Against the
    same launches with MCMC_SAB_CULL=0: chain states and every hit counter identical, the inelastic-only count
    identical, per-event products equal to rounding (only the order of the multiplications differs) -- central,
    peripheral, disjoint (b = 40 fm: product exactly 1, no hits) and asymmetric systems."""
    nN = nA + nB
    T, samples = 130, 5
    dmax2 = 6.76 / np.pi
    out = []
    for cull in ("1", "0"):
        option("MCMC_SAB_CULL", cull)
        e = _xs(dev, nA, nB, T, [PB] * nN, mode, seed=77)
        e.InitSampleFromPdf()
        e.WarmupMetropolis(6, stepsize=2.2)
        r1, m1, _ = e.MCIntegrate_S_AB(samples, b)                      # product only
        r1 = r1.cpu().numpy().copy()
        r3, m3, h3 = e.MCIntegrate_S_AB(samples, b, dmax2=dmax2)        # product + hit counters
        r3, h3 = r3.cpu().numpy().copy(), h3.cpu().numpy().copy()
        h4 = e.MCIntegrate_Inel(samples, b, dmax2).cpu().numpy().copy()  # inelastic flag only
        torch.cuda.synchronize()
        out.append((r1, r3, h3, h4, e.prevSample.cpu().numpy().copy()))
    (c1, c3, ch3, ch4, cs), (g1, g3, gh3, gh4, gs) = out
    assert np.array_equal(cs, gs)
    assert np.array_equal(ch3, gh3)
    assert np.array_equal(ch4, gh4)
    # a per-event mean is a sum of products of either sign (Gamma(0) > 1): rounding relative to the terms, not the sum
    np.testing.assert_allclose(c1, g1, rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(c3, g3, rtol=2e-5, atol=2e-7)
    assert np.array_equal(c1 == 0.0, g1 == 0.0)
    if b >= 40.0:
        assert np.all(c1 == 1.0) and ch3[0] == 0 and ch4[0] == 0
    if b == 13.0:
        assert 0.0 < c1.mean() < 1.0 and ch3[0] > 0


@pytest.mark.parametrize("nA,nB,b", [(208, 208, 9.0), (208, 208, 15.0), (12, 12, 2.0), (4, 4, 1.0)])
def test_sab_hit_counters_without_the_product(dev, nA, nB, b):
    """`mcmc_launch_integrate_sab` with a null resultBuffer (black-disk counters only, no exp): the counters and the
    chain advance equal those of the product + counters launch on the same draws."""
    from mcmc_amd import api, lib as L
    nN, T, samples = nA + nB, 200, 7
    dmax2 = 6.76 / np.pi
    tg = [PB] * nN if nN > 100 else [WS20] * nN
    out = []
    for product in (True, False):
        e = _xs(dev, nA, nB, T, tg, "recip_fast", seed=123)
        e.InitSampleFromPdf()
        e.WarmupMetropolis(5, stepsize=2.2)
        if product:
            _, _, h = e.MCIntegrate_S_AB(samples, b, dmax2=dmax2)
        else:
            L.check(L.lib.mcmc_launch_integrate_sab(C.byref(e.args), None, samples, C.c_float(b), None,
                                                    C.c_void_p(e.hits.data_ptr()), C.c_float(dmax2), api._stream()))
            h = e.hits
        torch.cuda.synchronize()
        out.append((h.cpu().numpy().copy(), e.prevSample.cpu().numpy().copy()))
    (h3, s3), (h2, s2) = out
    assert np.array_equal(h3, h2)
    assert np.array_equal(s3, s2)
    assert h3[3] == T * samples


@pytest.mark.parametrize("nucleus,A,c,a,r2", [("Au-197", 197, 6.38, 0.535, 28.377538), ("Pb-208", 208, 6.62, 0.546, 30.413841)])
def test_heavy_nucleus_r2_known_answer(dev, nucleus, A, c, a, r2):
    """BASELINE configs[2]: radial chains of every nucleon of Au-197 / Pb-208 (literature Woods-Saxon sets):
    warm-up + MCIntegrate_R2 with the width the device study finds (mcmc_tune_stepsize; the reference hard-codes 2.2),
    <r^2> against quadrature within 3 standard errors of the run-to-run scatter."""
    from mcmc_amd import api, lib as L
    best, _, _, tau = api.TuneStepSize(api.make_target(L.TARGET_R2WS, c, a), 2.0, 6.0, n_widths=9, n_chains=1 << 14,
                                       n_warmup=600, n_steps=1200)
    assert 3.0 < best < 5.0, best
    T, runs, samples = 1024, 8, 256
    xs = _xs(dev, A, 0, T, [(O.R2WS, c, a)] * A, "cross_fast", seed=1234)
    xs.InitSampleFromPdf()
    xs.WarmupMetropolis(512, stepsize=best)
    means = []
    for _ in range(runs):
        _, run_mean = xs.MCIntegrate_R2(samples)
        torch.cuda.synchronize()
        means.append(run_mean.cpu().numpy().mean())     # mean over nucleons of the per-nucleon run means
    means = np.array(means)
    est, se = means.mean(), means.std(ddof=1) / np.sqrt(runs)
    assert abs(est - r2) < 3 * se + 1e-3 * r2 * 0.0 + 3e-3, (nucleus, est, se, r2)
    assert se < 0.02


@pytest.mark.parametrize("nucleus,A,c,a,r2", [("Au-197", 197, 6.38, 0.535, 28.377538), ("Pb-208", 208, 6.62, 0.546, 30.413841)])
def test_heavy_nucleus_full_size_properties(dev, nucleus, A, c, a, r2):
    """BASELINE configs[2] AT ITS STATED SIZE: 2^22 coordinate chains (2^22 / A events x A nucleons) of the Au-197 / Pb-208
    Woods-Saxon density, through size-independent properties: slices of the event id space re-run as small engines with
    event_offset (the first events, an unaligned middle, the last ones) reproduce the big run's chain states and
    per-thread r^2 means bit for bit (what makes the 1/2/4/8-GPU partition exact); the device run means equal the FP32
    Kahan average of the result rows; <r^2> over all nucleons agrees with quadrature within 3 standard
    errors of the run-to-run scatter (width 3.9 fm: the device study's optimum for these densities)."""
    T = (1 << 22) // A
    tg = [(O.R2WS, c, a)] * A
    kw = dict(seed=20250611)
    big = _xs(dev, A, 0, T, tg, "cross_fast", **kw)
    big.InitSampleFromPdf()
    big.WarmupMetropolis(64, stepsize=3.9)
    start = big.prevSample.clone()
    res, run_mean = big.MCIntegrate_R2(128, stepsize=3.9)
    torch.cuda.synchronize()
    res_big, state_big, mean_big = res.clone(), big.prevSample.clone(), run_mean.cpu().numpy().copy()
    for off, cnt in ((0, 300), (T // 2 + 13, 517), (T - 77, 77)):
        part = _xs(dev, A, 0, cnt, tg, "cross_fast", event_offset=off, **kw)
        part.prevSample.copy_(start[:, off:off + cnt])
        part.args.step_offset = 64
        r, _ = part.MCIntegrate_R2(128, stepsize=3.9)
        torch.cuda.synchronize()
        assert torch.equal(part.prevSample, state_big[:, off:off + cnt]), off
        assert torch.equal(r, res_big[:, off:off + cnt]), off
    rows = res_big.cpu().numpy()
    for n in (0, A // 2, A - 1):
        assert mean_big[n] == pytest.approx(rows[n].astype(np.float64).mean(), rel=1e-12)
    means = [mean_big.mean()]
    for _ in range(7):
        _, rm = big.MCIntegrate_R2(128, stepsize=3.9)
        torch.cuda.synchronize()
        means.append(rm.cpu().numpy().mean())
    means = np.array(means)
    est, se = means.mean(), means.std(ddof=1) / np.sqrt(len(means))
    print(f"{nucleus}: <r^2> = {est:.5f} +/- {se:.5f} (quadrature {r2})")
    assert abs(est - r2) < 3 * se + 2e-4 * r2, (nucleus, est, se, r2)
    assert se < 5e-3


@pytest.mark.parametrize("kernel", ["warmup_r2", "sab", "inel"])
def test_log_space_rule_on_the_xsection_kernels(dev, kernel):
    """north_star subsystem (1), log-space acceptance, on the kernels that replace Metropolis.cuh:47-50: the LOG rule
    makes the same decisions as p' >= U p, so estimates agree with the NODIV rule on the same Philox draws within
    the statistical error (and the chains are identical wherever no proposal fell on a tie)."""
    T = 4096
    if kernel == "warmup_r2":
        out = {}
        for mode in ("log_fast", "nodiv_fast"):
            xs = _xs(dev, 6, 0, T, [WS20] * 6, mode, seed=8)
            xs.WarmupMetropolis(200, stepsize=2.2)
            _, rm = xs.MCIntegrate_R2(256)
            torch.cuda.synchronize()
            out[mode] = (rm.cpu().numpy().copy(), xs.prevSample.cpu().numpy().copy())
        same = (out["log_fast"][1] == out["nodiv_fast"][1]).mean()
        assert same > 0.97, same                      # 456 proposals per chain: a few chains meet a rounding tie
        np.testing.assert_allclose(out["log_fast"][0], out["nodiv_fast"][0], rtol=2e-3)
        assert abs(out["log_fast"][0].mean() - 6.424247) < 0.03
    else:
        out = {}
        for mode in ("log_fast", "nodiv_fast"):
            xs = _xs(dev, 12, 12, T, [WS20] * 24, mode, seed=9)
            xs.WarmupMetropolis(100, stepsize=2.2)
            if kernel == "sab":
                _, m, h = xs.MCIntegrate_S_AB(32, 2.5, dmax2=4.3 / np.pi)
                torch.cuda.synchronize()
                out[mode] = (m.item(), h.cpu().numpy().copy())
            else:
                h = xs.MCIntegrate_Inel(32, 4.0, 4.3 / np.pi)
                torch.cuda.synchronize()
                out[mode] = (0.0, h.cpu().numpy().copy())
        if kernel == "sab":
            assert out["log_fast"][0] == pytest.approx(out["nodiv_fast"][0], rel=0.02)
        np.testing.assert_allclose(out["log_fast"][1], out["nodiv_fast"][1], rtol=0.02)


def test_layer_b_run_all_writes_the_reference_tables(dev, tmp_path):
    """mcmc_config.outDir / outFile / integrateRMS / integrateSAB (main.cu:73-82) + mcmc_run_all (main.cu:157-416):
    the two tables with the TTree column names of IntegrationOutputHandler.cuh:83-112."""
    from mcmc_amd import lib as L
    cfg = L.Config()
    L.lib.mcmc_config_default(C.byref(cfg))
    assert cfg.integrateRMS == 1 and cfg.integrateSAB == 1
    cfg.totalBlocks, cfg.threadsPerBlock, cfg.samplesPerThread, cfg.totalRuns = 8, 128, 32, 3
    cfg.ndbs, cfg.db = 6, 1.0
    outdir = str(tmp_path).encode()
    cfg.outDir, cfg.outFile = outdir, b"he4"
    h = C.c_void_p()
    L.check(L.lib.mcmc_create(C.byref(cfg), C.byref(h)))
    try:
        L.check(L.lib.mcmc_warmup(h, 2.2, 128))
        sig = (C.c_double * 6)()
        L.check(L.lib.mcmc_run_all(h, sig))
    finally:
        L.check(L.lib.mcmc_destroy(h))
    rms = (tmp_path / "he4.RMS.csv").read_text().splitlines()
    sab = (tmp_path / "he4.S_AB.csv").read_text().splitlines()
    assert rms[0] == "Nucleus,Nucleon,RMS,RMS_stderr" and len(rms) == 1 + 8
    assert rms[1].startswith("1,1,") and rms[5].startswith("2,1,")
    assert sab[0] == "b,S_AB,S_AB_stderr" and len(sab) == 1 + 6
    assert sig[4] > 0 and sig[0] > sig[4]            # sigma_rxn > 0, sigma_tot > sigma_rxn
