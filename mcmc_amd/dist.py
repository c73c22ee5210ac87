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


def allreduce_stats(stats, n_lags=None, group=None):
    """Sum a [n_coords][MCMC_STATS_LEN] (or any) float64 statistics tensor over ranks, in place.
    A no-op when torch.distributed is not initialised (single GPU).  Every entry of the vector is a
    plain sum over chains (sums, sums of squares, accepts, lag products, head/tail sums, hit counts),
    so one SUM all-reduce is the whole exchange step."""
    if stats.dtype != torch.float64:
        raise TypeError("sufficient statistics are reduced in float64")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


def max_over_ranks(seconds, device=None, group=None):
    """Slowest rank's elapsed time (what whole-job throughput is quoted on)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
