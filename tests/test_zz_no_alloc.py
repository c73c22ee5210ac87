"""mcmc_chain_run_noalloc (the entry point for callers that cannot allocate: HIP graph capture, serving loops)
against mcmc_chain_run, through the C ABI: identical statistics and states, two calls in a row, the second with a
smaller input; oversize arguments are refused; a sequence of launches captured in a HIP graph and replayed equals
the same launches issued eagerly.  Collected last (test_zz_*) so that a failure here hides nothing."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _pair(dev, n_chains, n_coords, n_lags, seed, rule, math, precision=32):
    from mcmc_amd import api, lib as L
    kw = dict(n_coords=n_coords, target=api.make_target(L.TARGET_R2WS, 2.0, 0.54), step=2.48, precision=precision,
              accept_rule=rule, math_mode=math, n_lags=n_lags, center=2.3, seed=seed, chain_offset=12345, device=dev)
    return api.ChainEngine(n_chains, **kw), api.ChainEngine(n_chains, **kw)


def test_noalloc_matches_chain_run_twice_second_call_smaller(dev):
    from mcmc_amd import lib as L
    # call 1: ragged chain count over several grid-stride tiles, 49 lags (the largest statistics vector)
    # call 2: smaller in every dimension -- stale partials of call 1 remain in the scratch beyond what it uses
    for (nc, ncoord, nl, steps) in (((1 << 20) + 77, 3, 49, 96), (1000, 1, 0, 40)):
        ref, na = _pair(dev, nc, ncoord, nl, 99, L.ACCEPT_CROSS, L.MATH_FAST)
        s_ref = ref.run(steps).clone()
        s_na = na.run(steps, noalloc=True).clone()
        torch.cuda.synchronize()
        assert torch.equal(s_ref, s_na), (nc, ncoord, nl)
        assert torch.equal(ref.state, na.state)
        # a continuation launch (window carried on) behaves the same
        s_ref2 = ref.run(24, window_start=(nl == 0)).clone()
        s_na2 = na.run(24, window_start=(nl == 0), noalloc=True).clone()
        torch.cuda.synchronize()
        assert torch.equal(s_ref2, s_na2) and torch.equal(ref.state, na.state)
        assert float(s_na[0, 2]) > 0  # accepted proposals: the statistics are real


def test_noalloc_fp64_reference_rule_and_sample_writeback(dev):
    from mcmc_amd import lib as L
    ref, na = _pair(dev, 300, 2, 8, 5, L.ACCEPT_SHORTCIRCUIT, L.MATH_PRECISE, precision=64)
    smp_r = torch.empty((50, 2, 300), dtype=torch.float64, device=dev)
    smp_n = torch.empty_like(smp_r)
    a = ref.run(50, samples=smp_r).clone()
    b = na.run(50, samples=smp_n, noalloc=True).clone()
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(smp_r, smp_n)


def test_noalloc_rejects_beyond_the_scratch_size(dev):
    from mcmc_amd import api, lib as L
    # 2048 blocks x 24 coordinates x MCMC_STATS_LEN(49) doubles = 29.5 MB > MCMC_CHAIN_NOALLOC_MAX_BYTES (8 MiB)
    eng = api.ChainEngine(1 << 19, 24, api.make_target(L.TARGET_R2WS, 2.0, 0.54), 2.2, n_lags=49, device=dev)
    assert L.lib.mcmc_chain_workspace_bytes(C.byref(eng.args)) > L.CHAIN_NOALLOC_MAX_BYTES
    with pytest.raises(L.McmcError, match="MCMC_CHAIN_NOALLOC_MAX_BYTES"):
        eng.run(64, noalloc=True)
    eng.run(64)  # the workspace form still takes it
    torch.cuda.synchronize()


def test_noalloc_launches_captured_in_a_hip_graph(dev):
    from mcmc_amd import api, lib as L
    kw = dict(n_coords=2, target=api.make_target(L.TARGET_R2WS, 2.0, 0.54), step=2.48, precision=32,
              accept_rule=L.ACCEPT_CROSS, math_mode=L.MATH_FAST, n_lags=49, center=2.3, seed=7, chain_offset=99, device=dev)
    ref, na = api.ChainEngine(5000, **kw), api.ChainEngine(5000, **kw)
    # include/mcmc_hip.h: one chain launch on the device before the capture resolves the scratch address
    ref.run(64)
    na.run(64)
    torch.cuda.synchronize()
    before = na.state.clone()
    side = torch.cuda.Stream(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for i in range(3):
            na.run(64, window_start=(i == 0), noalloc=True)
    torch.cuda.synchronize()
    assert torch.equal(na.state, before)          # captured, not run
    graph.replay()
    for i in range(3):
        ref.run(64, window_start=(i == 0))
    torch.cuda.synchronize()
    assert torch.equal(ref.state, na.state)
    assert torch.equal(ref.stats, na.stats)
    assert float(na.stats[0, 2]) > 0
