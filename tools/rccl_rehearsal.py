#!/usr/bin/env python3
"""tools/rccl_rehearsal.py -- the collective calls of bench.py / mcmc_amd.dist on the RCCL backend with the
ranks a one-GPU box allows (one): process-group creation bound to the device, a float64 SUM all-reduce of a
statistics vector produced by the chain kernel on the kernel's stream, a MAX all-reduce, barrier, teardown.
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533
        tools/rccl_rehearsal.py"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import api, lib as L  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    lr = int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    dist.init_process_group("nccl", device_id=dev)
    eng = api.ChainEngine(1 << 16, 3, api.make_target(L.TARGET_R2WS, 2.0, 0.54), 2.48, precision=32,
                          accept_rule=L.ACCEPT_CROSS, math_mode=L.MATH_FAST, n_lags=0, chain_offset=rank << 16,
                          center=2.3, device=dev)
    st = eng.run(64, window_start=True)
    before = st.clone()
    dist.all_reduce(st, op=dist.ReduceOp.SUM)           # what mcmc_amd.dist.allreduce_stats issues for N > 1
    t = torch.tensor([1.25 + rank], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)            # mcmc_amd.dist.max_over_ranks
    dist.barrier()
    torch.cuda.synchronize()
    ok = bool(torch.equal(st, before)) if world == 1 else True
    if rank == 0:
        print(json.dumps({"backend": dist.get_backend(), "world": world, "sum_allreduce_f64_ok": ok,
                          "max_allreduce": t.item(), "accepts": float(st[0, 2].item())}))
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
