#!/usr/bin/env python3
"""tools/bench_mvn.py -- the stress configuration (BASELINE.json configs[4]) on one GPU:
2^21 chains on a 64-D Gaussian with Sigma_ij = 0.9^|i-j|, lag-1000 autocorrelation of x_0 through the
HBM ring, (a) observable write-back only, (b) full-state SoA write-back (the HBM-heavy path).
Prints one JSON object per variant with proposals/s and algorithmic HBM GB/s (peak 8 TB/s)."""
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
    ap.add_argument("--chains", type=int, default=1 << 21)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--lag", type=int, default=1000)
    ap.add_argument("--sweeps", type=int, default=64)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--step", type=float, default=0.8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    qd, qo = api.ar1_precision(args.dim, 0.9)
    eng = api.MvnEngine(args.chains, qd, qo, args.step, args.lag, math_mode=L.MATH_FAST, x0=0.0, device=dev)
    n, D, K = args.chains, args.dim, args.sweeps
    obs = torch.empty((K, n), dtype=torch.float32, device=dev)
    full = torch.empty((K, D, n), dtype=torch.float32, device=dev)
    big = (0, 1 << 40)
    eng.run(args.lag + 24, window=big)   # fill the lag ring: its slots are read from the window's sweep `lag` on
    variants = [("mvn_kernel", "0", "observable write-back + lag ring", dict(samples_obs=obs), 12),
                ("mvn_kernel", "0", "full-state SoA write-back + lag ring", dict(samples_obs=obs, samples_full=full), 12 + 4 * D)]
    if D == 64:   # the register-resident kernel exists for the stress configuration's own shape only
        variants.insert(0, ("mvn_reg_kernel", "1", "observable write-back + lag ring", dict(samples_obs=obs), 12))
        variants.insert(2, ("mvn_reg_kernel", "1", "full-state SoA write-back + lag ring", dict(samples_obs=obs, samples_full=full), 12 + 4 * D))
    for kernel, reg, name, kw, bytes_per in variants:
        L.check(L.lib.mcmc_set_option(b"MCMC_MVN_REG", reg.encode()))
        eng.run(K, window=big, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            eng.run(K, window=big, **kw)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = sorted(ts)[len(ts) // 2]
        # state load + store once per launch is part of the traffic too
        alg = n * (K * bytes_per + 2 * 4 * D)
        print(json.dumps({"kernel": kernel, "variant": name, "chains": n, "dim": D, "lag": args.lag, "sweeps": K,
                          "ms": ms, "proposals_per_s": n * D * K / (ms * 1e-3), "sweeps_per_s": n * K / (ms * 1e-3),
                          "algorithmic_bytes": alg, "hbm_GBps": alg / (ms * 1e-3) / 1e9, "hbm_frac_of_8TBps": alg / (ms * 1e-3) / 8e12,
                          # the other bound: 22 algorithmic FP32 flop per single-coordinate update against 157.3 TFLOP/s; the
                          # kernel is VALU-issue-bound (profiles/r2_valu_busy_calibrated.json), not HBM-bound as configs[4] assumes
                          "valu_TFLOPs": 22.0 * n * D * K / (ms * 1e-3) / 1e12, "valu_frac_of_157TFLOPs": 22.0 * n * D * K / (ms * 1e-3) / 157.3e12,
                          "bound": "valu issue"}))


if __name__ == "__main__":
    main()
