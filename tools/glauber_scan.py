#!/usr/bin/env python3
"""tools/glauber_scan.py -- BASELINE config 3 (C4) on N GPUs: the Pb-208 + Pb-208 impact-parameter scan with the
EVENTS sharded over ranks (every GPU contributes to every b) and one all-reduce of the [b][run][4] sufficient
statistics at the end (mcmc_amd.dist.glauber_scan / finalize_scan).
One GPU:   python tools/glauber_scan.py
N GPUs:    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
               tools/glauber_scan.py
Prints one JSON line on rank 0: sigma_inel (black disk), sigma_tot/el/rxn (unless --inel-only), event samples/s."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import api, dist as D, lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=1 << 14, help="events per launch, whole job")
    ap.add_argument("--samples", type=int, default=32)
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--nb", type=int, default=200)
    ap.add_argument("--b0", type=float, default=0.05)
    ap.add_argument("--db", type=float, default=0.1)
    ap.add_argument("--sigma-nn", type=float, default=6.76, help="fm^2 (67.6 mb)")
    ap.add_argument("--inel-only", action="store_true")
    ap.add_argument("--seed", type=int, default=314159265)
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("MCMC_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    off, cnt = D.partition(args.events, rank, world)
    pb = (L.TARGET_R2WS, 6.62, 0.546)   # literature Woods-Saxon of Pb-208 (not from the reference)
    step, _, _, _ = api.TuneStepSize(api.make_target(*pb), 2.0, 8.0, n_widths=13, seed=args.seed)
    eng = api.XsEngine(208, 208, cnt, [pb] * 416, stepsize=step, accept_rule=L.ACCEPT_CROSS, math_mode=L.MATH_FAST,
                       seed=args.seed, event_offset=off, sigmaNN=args.sigma_nn, device=dev)
    eng.InitSampleFromPdf()
    eng.WarmupMetropolis(256)
    b = args.b0 + args.db * np.arange(args.nb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = D.glauber_scan(eng, b, args.samples, args.runs, dmax2=args.sigma_nn / np.pi, inel_only=args.inel_only)
    D.allreduce_stats(stats)
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0, device=dev)
    if rank == 0:
        res = D.finalize_scan(stats.cpu().numpy(), b)
        out = {"system": "Pb-208 + Pb-208", "n_gpus": world, "events_per_launch": args.events, "samples": args.samples,
               "runs": args.runs, "impact_parameters": args.nb, "stepsize_tuned": step, "seconds": dt,
               "event_samples_per_s": args.events * args.samples * args.runs * args.nb / dt,
               "sigma_inel_black_disk_mb": res.get("sigma_inel_mb")}
        if "sigma" in res:
            out.update({"sigma_tot_mb": res["sigma"]["tot"], "sigma_el_mb": res["sigma"]["el"], "sigma_rxn_mb": res["sigma"]["rxn"]})
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
