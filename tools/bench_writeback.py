#!/usr/bin/env python3
"""tools/bench_writeback.py -- the chain kernel with every sample written back (2^24 chains x 3 coordinates x S sweeps; argv: S [case]), default
layout [S][C][n] (one 32-bit store per lane and step) against the pair layout [S/2][C][n][2] (option MCMC_SAMPLES_PAIRS: one
64-bit store per lane and step pair), and the same kernel without write-back.  One JSON line per case."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import api, lib as L  # noqa: E402

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
only = sys.argv[2] if len(sys.argv) > 2 else None
nc, C = 1 << 24, 3
reps = 5


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2] * 1e-3


eng = api.ChainEngine(nc, C, api.make_target(L.TARGET_R2WS, 2.0, 0.54), 2.48, precision=32, accept_rule=L.ACCEPT_CROSS,
                      math_mode=L.MATH_FAST, center=2.3, device=dev)
smp = torch.empty((S, C, nc), dtype=torch.float32, device=dev)
for name in ("none", "rows", "pairs"):
    if only and name != only:
        continue
    if name == "pairs":
        L.check(L.lib.mcmc_set_option(b"MCMC_SAMPLES_PAIRS", b"1"))
    t = timed(lambda: eng.run(S, samples=None if name == "none" else smp))
    L.lib.mcmc_set_option(b"MCMC_SAMPLES_PAIRS", None)
    byts = (nc * C * S * 4 if name != "none" else 0) + 2 * 4 * nc * C
    print(json.dumps({"layout": name, "chains": nc, "coords": C, "sweeps": S, "ms": t * 1e3, "proposals_per_s": nc * C * S / t,
                      "algorithmic_GBps": byts / t / 1e9}))
