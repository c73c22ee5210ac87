#!/usr/bin/env python3
"""tools/bench_store.py -- the chain kernel WITH coalesced SoA sample write-back (north_star: "HBM GB/s when
samples are stored"): 2^24 chains x 3 coordinates, every state written to samples[step][coord][chain].
Algorithmic traffic = 4 B per proposal (+ state load/store once per launch); HBM peak 8 TB/s."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import api, lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=1 << 24)
    ap.add_argument("--coords", type=int, default=3)
    ap.add_argument("--sweeps", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    eng = api.ChainEngine(args.chains, args.coords, api.make_target(L.TARGET_R2WS, 2.0, 0.54), 2.48, precision=32,
                          accept_rule=L.ACCEPT_CROSS, math_mode=L.MATH_FAST, center=2.3, device=dev)
    samples = torch.empty((args.sweeps, args.coords, args.chains), dtype=torch.float32, device=dev)
    for name, kw in (("no write-back", {}), ("samples write-back", {"samples": samples})):
        eng.run(args.sweeps, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            eng.run(args.sweeps, **kw)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = sorted(ts)[len(ts) // 2]
        props = args.chains * args.coords * args.sweeps
        alg = (props * 4 if kw else 0) + 2 * 4 * args.chains * args.coords
        print(json.dumps({"kernel": "chain_kernel", "variant": name, "chains": args.chains, "coords": args.coords,
                          "sweeps": args.sweeps, "ms": ms, "proposals_per_s": props / (ms * 1e-3), "algorithmic_bytes": alg,
                          "hbm_GBps": alg / (ms * 1e-3) / 1e9, "hbm_frac_of_8TBps": alg / (ms * 1e-3) / 8e12}))


if __name__ == "__main__":
    main()
