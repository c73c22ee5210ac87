"""N > 1 path on CPU: two `gloo` ranks shard the chains by global id, reduce the
sufficient statistics with ONE all-reduce (mcmc_amd.dist) and finalise them with
the C ABI; the result must equal the single-process answer over all chains.

No GPU here, so each rank's statistics vector is produced by the CPU oracle's
restatement of the chain (the checker standing in for the kernel); what is under
test is the product's host logic: partitioning, the collective, finalisation.
The GPU twin of this test is test_gpu_chain.py::test_split_launches_and_chain_
partition_are_invariant (chain_offset on a real device)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_CHAINS, SWEEPS, SEED = 1001, 50, 2024


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_stats(chain0, count):
    import oracle_lib as O
    orc = O.Oracle()
    acc, s1, s2 = orc.baseline_chains_f32(chain0, count, 1, O.R2WS, 2.0, 0.54, 2.2, SEED, 0, SWEEPS)
    return np.array([[s1, s2, float(acc)]])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from mcmc_amd import dist as D
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        off, cnt = D.partition(N_CHAINS, rank, world)
        st = torch.from_numpy(_rank_stats(off, cnt))
        D.allreduce_stats(st)
        t = D.max_over_ranks(1.0 + rank)
        q.put((rank, off, cnt, st.numpy().copy(), t))
    finally:
        dist.destroy_process_group()


def test_partition_covers_ids_exactly_once():
    from mcmc_amd import dist as D
    for n, w in ((1001, 2), (16, 8), (7, 8), (1 << 24, 8), (5, 1)):
        spans = [D.partition(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (o0, c0), (o1, _) in zip(spans, spans[1:]):
            assert o0 + c0 == o1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        D.partition(10, 2, 2)


def test_two_rank_gloo_allreduce_equals_single_process():
    from mcmc_amd import api
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = _rank_stats(0, N_CHAINS)
    assert res[0][1:3] == (0, 501) and res[1][1:3] == (501, 500)
    for _, _, _, st, t in res:
        np.testing.assert_allclose(st, whole, rtol=1e-13)   # same chains, summed in two halves
        assert t == 2.0                                     # MAX over ranks
    out = api.finalize_stats(res[0][3][0], 0, N_CHAINS, SWEEPS, 0.0)
    ref = api.finalize_stats(whole[0], 0, N_CHAINS, SWEEPS, 0.0)
    assert out["mean"] == pytest.approx(ref["mean"], rel=1e-13)
    assert out["acceptance"] == ref["acceptance"]


def _scan_rank_stats(rank, nb=7, runs=3):
    """Synthetic per-rank contributions of the Glauber scan: [b][run][sum S_AB, events, inelastic samples, samples]."""
    rng = np.random.default_rng(100 + rank)
    events = 500 + rank
    st = np.zeros((nb, runs, 4))
    s_ab = np.linspace(0.0, 1.0, nb)[:, None] + 0.01 * rng.standard_normal((nb, runs))
    st[:, :, 0] = s_ab * events
    st[:, :, 1] = events
    st[:, :, 3] = events * 16
    st[:, :, 2] = np.floor(st[:, :, 3] * np.linspace(1.0, 0.0, nb)[:, None])
    return st


def _scan_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mcmc_amd import dist as D
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        st = torch.from_numpy(_scan_rank_stats(rank))
        D.allreduce_stats(st)
        res = D.finalize_scan(st.numpy(), np.arange(7) * 2.0)
        q.put((rank, res["S_AB"].copy(), res["S_AB_stderr"].copy(), res["p_inel"].copy(), res["sigma_inel_mb"],
               res["sigma"]["rxn"][0]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_glauber_scan_reduction():
    """The C4 sharding: per-rank [b][run][4] tensors, ONE all-reduce, the reference's run estimator (main.cu:359,
    384-386) and the cross sections evaluated on the totals == the same evaluation of the directly summed tensors."""
    from mcmc_amd import dist as D
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scan_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = D.finalize_scan(_scan_rank_stats(0) + _scan_rank_stats(1), np.arange(7) * 2.0)
    for _, s, e, pin, sig_inel, sig_rxn in res:
        np.testing.assert_allclose(s, whole["S_AB"], rtol=1e-7)
        np.testing.assert_allclose(e, whole["S_AB_stderr"], rtol=1e-5)
        np.testing.assert_allclose(pin, whole["p_inel"], rtol=1e-14)
        assert sig_inel == pytest.approx(whole["sigma_inel_mb"], rel=1e-13)
        assert sig_rxn == pytest.approx(whole["sigma"]["rxn"][0], rel=1e-6)
    assert whole["S_AB"][0] < 0.05 and whole["S_AB"][-1] > 0.95 and np.all(whole["S_AB_stderr"] > 0)
    assert whole["p_inel"][0] == 1.0 and whole["p_inel"][-1] == 0.0
