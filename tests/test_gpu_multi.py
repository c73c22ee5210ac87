"""Layer C of the C ABI (include/mcmc_hip.h): the drivers' hot paths partitioned over several GPUs of one node, one
all-reduce of sufficient statistics per batch (north_star; reference analogue of the reduction: the host Average over
threads and runs, xsection/src/main.cu:221-223,253-258,359,384-389).

The box has ONE GPU: several shards on device 0 rehearse the partition with the host reduction (what a node without
librccl runs), and the RCCL path runs with one rank.  xsection payloads are integers (hit counts, fixed-point sums),
so every result must be bit-identical for any number of shards."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPSIZE = os.path.join(ROOT, "mcmc_amd", "host", "stepsize")
XSECTION = os.path.join(ROOT, "mcmc_amd", "host", "xsection")


def _cfg(nucleons=2, blocks=9, threads=100, samples=16, runs=3, ndbs=5, db=0.9, seed=77):
    from mcmc_amd import lib as L
    cfg = L.Config()
    L.lib.mcmc_config_default(C.byref(cfg))
    cfg.nNeutronA = cfg.nProtonA = cfg.nNeutronB = cfg.nProtonB = nucleons
    cfg.totalBlocks, cfg.threadsPerBlock, cfg.samplesPerThread, cfg.totalRuns = blocks, threads, samples, runs
    cfg.ndbs, cfg.db, cfg.seed = ndbs, db, seed
    cfg.accept_rule, cfg.math_mode = L.ACCEPT_CROSS, L.MATH_FAST
    return cfg


def _run_multi(cfg, devices, reduce, random_start=True):
    """warm-up, RMS integration, S_AB scan with black-disk counters, inelastic-only scan -> every output + the states."""
    from mcmc_amd import lib as L
    m = C.c_void_p()
    dv = (C.c_int * len(devices))(*devices)
    L.check(L.lib.mcmc_multi_create(C.byref(cfg), dv, len(devices), reduce, C.byref(m)))
    try:
        ns, used = C.c_int(), C.c_int()
        off, cnt = (C.c_uint64 * len(devices))(), (C.c_uint64 * len(devices))()
        L.check(L.lib.mcmc_multi_info(m, C.byref(ns), C.byref(used), off, cnt))
        assert ns.value == len(devices)
        T = cfg.totalBlocks * cfg.threadsPerBlock
        assert sum(cnt[:]) == T and off[0] == 0
        for r in range(1, len(devices)):
            assert off[r] == off[r - 1] + cnt[r - 1] and 0 <= cnt[r - 1] - cnt[r] <= 1   # contiguous, balanced
        if random_start:
            L.check(L.lib.mcmc_multi_init_from_pdf(m))
        L.check(L.lib.mcmc_multi_warmup(m, 2.2, 40))
        nN = cfg.nNeutronA + cfg.nProtonA + cfg.nNeutronB + cfg.nProtonB
        rms, err = (C.c_float * nN)(), (C.c_float * nN)()
        L.check(L.lib.mcmc_multi_run_rms(m, rms, err))
        n = cfg.ndbs
        b, S, E, pin = (C.c_float * n)(), (C.c_float * n)(), (C.c_float * n)(), (C.c_double * n)()
        L.check(L.lib.mcmc_multi_run_sab_scan(m, b, S, E, pin, 4.3 / np.pi))
        b2, pin2, perr2 = (C.c_float * n)(), (C.c_double * n)(), (C.c_double * n)()
        L.check(L.lib.mcmc_multi_run_inel_scan(m, b2, pin2, perr2, 4.3 / np.pi))
        states = []
        for r in range(len(devices)):
            h = L.lib.mcmc_multi_shard(m, r)
            assert h
            st = np.empty((nN, cnt[r]), dtype=np.float32)
            L.check(L.lib.mcmc_get_state(h, st.ctypes.data_as(C.POINTER(C.c_float))))
            states.append(st)
        out = dict(rms=np.array(rms[:]), err=np.array(err[:]), b=np.array(b[:]), S=np.array(S[:]), E=np.array(E[:]),
                   pin=np.array(pin[:]), pin2=np.array(pin2[:]), perr2=np.array(perr2[:]), state=np.concatenate(states, axis=1),
                   used=used.value)
    finally:
        L.check(L.lib.mcmc_multi_destroy(m))
    return out


@pytest.mark.parametrize("nucleons", [2, 6])   # He-4 + He-4 (main.cu:38-41) and C-12 + C-12
def test_shards_on_one_device_equal_one_shard_bit_for_bit(dev, nucleons):
    """1, 2 and 3 shards (900 events: 3 shards of 300, 2 of 450) through the host reduction: RMS, S_AB, its standard
    error, both inelastic probabilities and the chain states identical to the last bit."""
    from mcmc_amd import lib as L
    cfg = _cfg(nucleons=nucleons)
    one = _run_multi(cfg, [0], L.REDUCE_HOST)
    assert one["used"] == L.REDUCE_HOST
    assert 0.0 < one["S"][0] < one["S"][-1] <= 1.0 and one["pin"][0] > one["pin"][-1] and np.all(one["E"] > 0)
    assert np.all(np.abs(one["rms"] - 2.534610) < 0.2)
    assert np.all(np.abs(one["pin"] - one["pin2"]) < 0.05)   # the flag-only scan runs on later sweeps of the same chains
    for devices in ([0, 0], [0, 0, 0], [0] * 7):
        many = _run_multi(cfg, devices, L.REDUCE_AUTO)       # two shards share a device: AUTO must fall back to the host
        assert many["used"] == L.REDUCE_HOST
        for key in ("rms", "err", "b", "S", "E", "pin", "pin2", "perr2", "state"):
            np.testing.assert_array_equal(many[key], one[key], err_msg=f"{key} with {len(devices)} shards")


def test_rccl_all_reduce_with_one_rank_and_refusals(dev):
    """The RCCL path (ncclCommInitAll + grouped ncclAllReduce on the shard's stream) with the one rank this box has:
    same results as the host reduction.  RCCL with two shards on one device is refused with a message."""
    from mcmc_amd import lib as L
    cfg = _cfg()
    host = _run_multi(cfg, [0], L.REDUCE_HOST)
    rccl = _run_multi(cfg, [0], L.REDUCE_RCCL)
    assert rccl["used"] == L.REDUCE_RCCL
    for key in ("rms", "err", "S", "E", "pin", "pin2", "state"):
        np.testing.assert_array_equal(rccl[key], host[key], err_msg=key)
    m = C.c_void_p()
    dv = (C.c_int * 2)(0, 0)
    assert L.lib.mcmc_multi_create(C.byref(cfg), dv, 2, L.REDUCE_RCCL, C.byref(m)) == -1
    assert b"share a device" in L.lib.mcmc_last_error()
    dv = (C.c_int * 1)(99)
    assert L.lib.mcmc_multi_create(C.byref(cfg), dv, 1, L.REDUCE_HOST, C.byref(m)) == -1
    assert b"out of range" in L.lib.mcmc_last_error()


def test_sharded_job_agrees_with_the_one_gpu_handle(dev):
    """mcmc_multi_* against mcmc_create / mcmc_run_*: the same events, draws and estimator; only the sum over threads
    differs (fixed-point integers against the FP64 tree): counts exactly, S_AB and RMS to a float ulp."""
    from mcmc_amd import lib as L
    cfg = _cfg(blocks=8, threads=128)
    multi = _run_multi(cfg, [0, 0], L.REDUCE_HOST, random_start=False)
    h = C.c_void_p()
    L.check(L.lib.mcmc_create(C.byref(cfg), C.byref(h)))
    try:
        L.check(L.lib.mcmc_warmup(h, 2.2, 40))
        rms, err = (C.c_float * 8)(), (C.c_float * 8)()
        L.check(L.lib.mcmc_run_rms(h, rms, err))
        n = cfg.ndbs
        b, S, E, pin = (C.c_float * n)(), (C.c_float * n)(), (C.c_float * n)(), (C.c_double * n)()
        L.check(L.lib.mcmc_run_sab_scan(h, b, S, E, pin, 4.3 / np.pi))
    finally:
        L.check(L.lib.mcmc_destroy(h))
    np.testing.assert_array_equal(multi["pin"], np.array(pin[:]))
    np.testing.assert_allclose(multi["S"], np.array(S[:]), rtol=2.5e-7)
    np.testing.assert_allclose(multi["rms"], np.array(rms[:]), rtol=2.5e-7)
    np.testing.assert_allclose(multi["E"], np.array(E[:]), rtol=1e-3, atol=1e-9)


def test_fixed_point_row_sum_is_exact_and_partition_invariant(dev):
    """mcmc_launch_row_sum_fixed_f32: the limb sums equal the exact integer sum of round(x 2^40) (Python integers), for
    one launch over all columns and for column slices added up; entries that are not finite or too large are counted."""
    from mcmc_amd import lib as L
    rng = np.random.default_rng(3)
    rows, cols = 3, 100003
    x = (rng.standard_normal((rows, cols)) * np.array([[1e-3], [1.0], [300.0]])).astype(np.float32)
    x[1, 17] = 0.0
    x[2, 5] = 16383.9
    xd = torch.from_numpy(x).to(dev)
    ws_bytes = L.lib.mcmc_row_sum_fixed_workspace_bytes(rows)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out = torch.zeros((rows, 4), dtype=torch.float64, device=dev)
    L.check(L.lib.mcmc_launch_row_sum_fixed_f32(C.c_void_p(xd.data_ptr()), rows, cols, C.c_void_p(out.data_ptr()),
                                                C.c_void_p(ws.data_ptr()), ws_bytes, None))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    q = np.rint(x.astype(np.float64) * 2.0 ** 40).astype(np.int64)   # |x| < 2^14: exact in float64 and int64
    for r in range(rows):
        exact = sum(int(v) for v in q[r])
        got = (int(o[r, 0]) << 40) + (int(o[r, 1]) << 20) + int(o[r, 2])
        assert got == exact and o[r, 3] == 0
        limbs = (C.c_double * 3)(*o[r, :3])
        assert L.lib.mcmc_fixed_sum_value(limbs) == float(exact) * 2.0 ** -40 or \
            abs(L.lib.mcmc_fixed_sum_value(limbs) - exact / 2 ** 40) <= abs(exact / 2 ** 40) * 2.0 ** -52
    # column slices of row 1, each a launch of its own (what shards do), add up to the same integers
    tot = np.zeros(4)
    for lo, hi in ((0, 1), (1, 40000), (40000, cols)):
        part = xd[1, lo:hi].contiguous()
        L.check(L.lib.mcmc_launch_row_sum_fixed_f32(C.c_void_p(part.data_ptr()), 1, hi - lo, C.c_void_p(out.data_ptr()),
                                                    C.c_void_p(ws.data_ptr()), ws_bytes, None))
        torch.cuda.synchronize()
        tot += out.cpu().numpy()[0]
    np.testing.assert_array_equal(tot, o[1])
    bad = xd.clone()
    bad[0, 3], bad[0, 4], bad[0, 9] = float("nan"), float("inf"), 20000.0
    L.check(L.lib.mcmc_launch_row_sum_fixed_f32(C.c_void_p(bad.data_ptr()), rows, cols, C.c_void_p(out.data_ptr()),
                                                C.c_void_p(ws.data_ptr()), ws_bytes, None))
    torch.cuda.synchronize()
    assert out.cpu().numpy()[0, 3] == 3


@pytest.mark.parametrize("precision,rule", [(32, "cross"), (64, "shortcircuit")])
def test_stepsize_run_sharded_equals_one_shard(dev, precision, rule):
    """mcmc_stepsize_run_multi: the chains as 1, 2 and 3 shards (chain_offset = the shard's first global id).
    Acceptance is a ratio of integer counts: exact.  Means and variances are FP64 sums over chains in another grouping:
    equal to FP64 rounding.  The lag products: FP64 (per chain) in the 64-bit kernel -- equal to FP64 rounding; in the
    32-bit kernel they are pooled per wavefront of 64 chains in FP32 tiles on the matrix cores (chain_lag_kernel.hpp),
    so they are equal to FP64 rounding when every shard starts on a multiple of 64 chains (1024 chains in 2 shards:
    the same chains share a wavefront) and to FP32 accumulation rounding otherwise (1000 chains; 3 shards)."""
    from mcmc_amd import lib as L
    tgt = L.Target(L.TARGET_R2WS, 2.0, 0.54)
    acc = {"cross": L.ACCEPT_CROSS, "shortcircuit": L.ACCEPT_SHORTCIRCUIT}[rule]
    math = L.MATH_FAST if precision == 32 else L.MATH_PRECISE
    for n_chains, shard_lists in ((1000, ([0], [0, 0], [0, 0, 0])), (1024, ([0], [0, 0]))):
        res = []
        for devices in shard_lists:
            dv = (C.c_int * len(devices))(*devices)
            summ = (L.ChainSummary * 2)()
            rho = (C.c_double * (2 * 49))()
            L.check(L.lib.mcmc_stepsize_run_multi(C.byref(tgt), precision, acc, math, 2.48, n_chains, 2, 50, 400, 49, 12345, 1.0,
                                                  dv, len(devices), L.REDUCE_AUTO, summ, rho, None))
            res.append(([(s.mean, s.variance, s.acceptance, s.tau_int) for s in summ], np.array(rho[:])))
        exact_lags = precision == 64 or n_chains % 128 == 0
        for other in res[1:]:
            for c in range(2):
                assert other[0][c][2] == res[0][0][c][2]
                np.testing.assert_allclose(other[0][c][:2], res[0][0][c][:2], rtol=1e-11)
                assert other[0][c][3] == pytest.approx(res[0][0][c][3], rel=1e-11 if exact_lags else 2e-5)
            np.testing.assert_allclose(other[1], res[0][1], rtol=0, atol=1e-11 if exact_lags else 3e-6)
        assert 0.4 < res[0][0][0][2] < 0.6 and 2.0 < res[0][0][0][3] < 8.0


def test_stepsize_run_fp32_long_window_is_split(dev, oracle):
    """stepsize/src/main.cpp:16-30 at the reference's own size -- ONE chain, 10^6 iterations, lags 1..49 -- in FP32.
    The library cuts the window into ring-carried launches of at most 65536 steps (their statistics added in FP64) and
    centres the accumulators on the target's mean, so the streaming rho_k equal statistics.hpp:36-57 evaluated in FP64
    on the very samples the run wrote back, to 3e-5 (the bar of the FP64 golden test)."""
    from mcmc_amd import lib as L
    n = 1_000_000
    tgt = L.Target(L.TARGET_R2WS, 2.0, 0.54)
    summ = (L.ChainSummary * 1)()
    rho = (C.c_double * 49)()
    samples = np.empty(n, dtype=np.float32)
    L.check(L.lib.mcmc_stepsize_run(C.byref(tgt), 32, L.ACCEPT_RECIP, L.MATH_FAST, 2.48, 1, 1, 0, n, 49, 2024, 1.0, summ, rho,
                                    samples.ctypes.data_as(C.c_void_p)))
    x = samples.astype(np.float64)
    assert summ[0].mean == pytest.approx(x.mean(), rel=1e-6)
    assert summ[0].variance == pytest.approx(x.var(ddof=1), rel=1e-5)
    worst = 0.0
    for k in range(1, 50):
        ref = oracle.autocorrelation_f64(x, k)
        worst = max(worst, abs(rho[k - 1] - ref))
    print(f"largest |rho_k - reference formula| over k = 1..49: {worst:.2e}")
    assert worst < 3e-5
    assert abs(summ[0].mean - 2.3317) < 0.02


def test_drivers_on_several_shards(dev, tmp_path):
    """xsection devices=0,0,0 and stepsize --devices 0,0: the drivers' output files equal the one-shard files byte for
    byte (xsection) / to the printed digits (stepsize)."""
    common = ["totalBlocks=10", "threadsPerBlock=100", "samplesPerThread=32", "totalRuns=3", "samplesForMix=200", "ndbs=8",
              "db=0.8", "blackDisk=1", "rule=cross", "randomStart=1", f"outDir={tmp_path}"]
    for tag, devs in (("one", "0"), ("three", "0,0,0")):
        r = subprocess.run([XSECTION, *common, f"devices={devs}", "reduce=host", f"outFile={tag}"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "statistics reduced by host sum" in r.stdout
    for ext in ("RMS.csv", "S_AB.csv"):
        assert (tmp_path / f"one.{ext}").read_bytes() == (tmp_path / f"three.{ext}").read_bytes()
    r = subprocess.run([XSECTION, *common, "devices=0", "reduce=rccl", "outFile=rccl"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "RCCL all-reduce" in r.stdout
    assert (tmp_path / "one.S_AB.csv").read_bytes() == (tmp_path / "rccl.S_AB.csv").read_bytes()
    rows = []
    for devs in ("0", "0,0"):
        d = tmp_path / f"s{len(devs)}"
        d.mkdir()
        r = subprocess.run([STEPSIZE, "--chains", "1024", "--iterations", "2000", "--first", "248", "--last", "248", "--precision",
                            "32", "--rule", "cross", "--math", "fast", "--devices", devs, "--outdir", str(d)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr + r.stdout
        rows.append(np.loadtxt(d / "statistics.txt", comments="#"))
    np.testing.assert_allclose(rows[0], rows[1], rtol=1e-6, atol=1e-9)
