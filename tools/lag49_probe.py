#!/usr/bin/env python3
"""tools/lag49_probe.py -- the 49-lag chain kernel alone: proposals/s over launches of `--sweeps` sweeps and the shader
clock it sustains (mcmc_chain_clock_probe), for rocprofv3 passes and A/B timing."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import api, lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=1 << 22)
ap.add_argument("--sweeps", type=int, default=1024)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--lags", type=int, default=49)
ap.add_argument("--no-clock", action="store_true")
ap.add_argument("--batch", type=int, default=1, help="launches between the two events (back to back, one window each)")
a = ap.parse_args()
dev = torch.device("cuda:0")
e = api.ChainEngine(a.chains, 3, api.make_target(L.TARGET_R2WS, 2.0, 0.54), 2.48, precision=32, accept_rule=L.ACCEPT_CROSS,
                    math_mode=L.MATH_FAST, n_lags=a.lags, center=2.3, device=dev)
e.run(256, n_lags=a.lags)
torch.cuda.synchronize()
ts = []
for _ in range(a.reps):
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.batch):
        e.run(a.sweeps, n_lags=a.lags)
    t.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(t) / a.batch)
ms = sorted(ts)[len(ts) // 2]
ghz = None if a.no_clock else api.chain_clock_probe(e, a.sweeps, n_lags=a.lags)
props = a.chains * 3 * a.sweeps
out = {"kernel": f"chain_kernel lags={a.lags}", "chains": a.chains, "sweeps": a.sweeps, "batch": a.batch, "ms": ms, "proposals_per_s": props / (ms * 1e-3),
       "clock_GHz": ghz, "tau_int": e.summary()[0]["tau_int"] if a.lags else None}
if ghz:
    out["simd_cycles_per_wave_proposal"] = 1024 * ghz * 1e9 * 64 * (ms * 1e-3) / props
print(json.dumps(out))
