#!/usr/bin/env python3
"""tools/bench_f64.py -- the FP64 stepsize path (metropolis.hpp arithmetic, rule SHORTCIRCUIT) on one GPU: proposals/s of
chain_kernel<double> without and with the 49 lag products (the reference's C1 / C2 arithmetic at device scale)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import api, lib as L  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    tg = api.make_target(L.TARGET_R2WS, 2.0, 0.54)
    for n_chains, lags, steps in ((1 << 20, 0, 256), (1 << 20, L.MAX_LAGS, 256), (1 << 14, L.MAX_LAGS, 1024)):
        eng = api.ChainEngine(n_chains, 1, tg, 2.48, precision=64, accept_rule=L.ACCEPT_SHORTCIRCUIT, math_mode=L.MATH_PRECISE,
                              n_lags=L.MAX_LAGS if lags else 0, seed=1, x0=1.0, center=2.3, device=dev)
        eng.run(steps, n_lags=lags, window_start=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            eng.run(steps, n_lags=lags, window_start=True)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = sorted(ts)[len(ts) // 2]
        print(json.dumps({"kernel": "chain_kernel<double, SHORTCIRCUIT, PRECISE>", "chains": n_chains, "lags": lags, "steps": steps,
                          "ms": ms, "proposals_per_s": n_chains * steps / (ms * 1e-3)}))


if __name__ == "__main__":
    main()
