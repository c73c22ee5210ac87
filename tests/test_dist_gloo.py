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
