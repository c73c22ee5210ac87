"""Multi-GPU sharding of the chain path: one process per GPU, chains partitioned
by global chain id, ONE all-reduce of the sufficient-statistics vector per batch
(RCCL over xGMI on the GPU box, `gloo` in the CPU tests).

The reference has no multi-GPU path; its analogue of the reduction is the host
`Average` over threads and then over runs (xsection/src/main.cu:221-223,254).
Because the Philox counter carries the GLOBAL chain id (mcmc_chain_args.
chain_offset / mcmc_xs_args.event_offset), the union of the ranks' chains is
exactly the chain set a single device would run, so summed statistics do not
depend on the number of ranks.
"""
import torch
import torch.distributed as dist


def partition(n_total, rank, world):
    """Contiguous, balanced partition of ids [0, n_total): -> (offset, count) of `rank`.
    The first n_total % world ranks own one extra id."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} for world size {world}")
    base, extra = divmod(int(n_total), int(world))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def _all_reduce(t, op, group=None):
    """all_reduce of a device tensor; `gloo` (CPU tests, one-GPU rehearsals) gets a host copy, RCCL the tensor itself."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def allreduce_stats(stats, n_lags=None, group=None):
    """Sum a [n_coords][MCMC_STATS_LEN] (or any) float64 statistics tensor over ranks, in place.
    A no-op when torch.distributed is not initialised (single GPU).  Every entry of the vector is a
    plain sum over chains (sums, sums of squares, accepts, lag products, head/tail sums, hit counts),
    so one SUM all-reduce is the whole exchange step."""
    if stats.dtype != torch.float64:
        raise TypeError("sufficient statistics are reduced in float64")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        _all_reduce(stats, dist.ReduceOp.SUM, group)
    return stats


def max_over_ranks(seconds, device=None, group=None):
    """Slowest rank's elapsed time (what whole-job throughput is quoted on)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    _all_reduce(t, dist.ReduceOp.MAX, group)
    return float(t.item())


# --- the Glauber impact-parameter scan, events sharded over ranks -------------------
# Rank r owns the events [offset, offset + count) of the global id space (XsEngine(event_offset=offset)),
# runs EVERY impact parameter on them, and writes four numbers per (b, run) into one device tensor;
# ONE all-reduce(SUM) of that tensor is the whole exchange.  The reference's estimator (main.cu:359,
# 384-389: mean over threads per run, then Average / Variance over runs) is evaluated on the totals.
SCAN_COLS = 4  # sum over events of the per-event S_AB mean | events | samples with >= 1 collision | samples


def glauber_scan(engine, b_values, samples, runs, dmax2=None, inel_only=False):
    """All launches of the scan queued on the current stream; returns the [len(b)][runs][SCAN_COLS] float64
    device tensor of THIS rank's contributions (no host synchronisation)."""
    stats = torch.zeros((len(b_values), int(runs), SCAN_COLS), dtype=torch.float64, device=engine.device)
    for i, b in enumerate(b_values):
        for r in range(int(runs)):
            row = stats[i, r]
            if inel_only:
                hits = engine.MCIntegrate_Inel(samples, float(b), dmax2)
                row[2:4] = hits[0:2]
            else:
                _, mean, hits = engine.MCIntegrate_S_AB(samples, float(b), dmax2=dmax2)
                row[0] = mean[0] * engine.T
                if hits is not None:
                    row[2] = hits[1]
                    row[3] = hits[3]
            row[1] = engine.T
    return stats


def finalize_scan(stats_total, b_values):
    """Host side, on the all-reduced tensor: per b the reference's run estimator of S_AB (FP32 Kahan Average /
    Variance through the C ABI, main.cu:384-386), the black-disk inelastic probability, and the cross sections
    (XSectionAnalysis.py:3-28 via mcmc_cross_sections; sigma_inel by the same trapezoid rule)."""
    import numpy as np
    from . import api
    st = np.asarray(stats_total, dtype=np.float64)
    b = np.asarray(b_values, dtype=np.float64)
    nb, runs, _ = st.shape
    out = {"b": b, "S_AB": np.zeros(nb), "S_AB_stderr": np.zeros(nb), "p_inel": np.zeros(nb)}
    for i in range(nb):
        run_means = (st[i, :, 0] / np.maximum(st[i, :, 1], 1.0)).astype(np.float32)   # main.cu:359
        avg = api.Average(run_means)
        out["S_AB"][i] = avg
        out["S_AB_stderr"][i] = np.sqrt(api.Variance(run_means, avg) / runs) if runs > 1 else 0.0
        tot = st[i, :, 3].sum()
        out["p_inel"][i] = st[i, :, 2].sum() / tot if tot > 0 else 0.0
    if nb >= 2:
        if st[:, :, 0].any():
            out["sigma"] = api.CrossSections(b, out["S_AB"], out["S_AB_stderr"])
        if st[:, :, 3].any():
            w = b * out["p_inel"]
            out["sigma_inel_mb"] = 10.0 * 2.0 * np.pi * float(np.sum(0.5 * (w[1:] + w[:-1]) * np.diff(b)))
    return out
