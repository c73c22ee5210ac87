#!/usr/bin/env python3
"""tools/known_answer_large.py -- the chain kernel against quadrature at bench scale: <r> and <r^2> of r^2 WS(2.0, 0.54)
from 2^24 chains x 3 coordinates x (512 warm-up + 4096 measured) sweeps (2.1e11 samples per rule), the statistical error
from the spread over 64 chain blocks.  A check of the packed 23-bit draws (the fields' low bits, the top-byte spares) at a
resolution the unit tests do not reach.  Quadrature: scipy, independent of any reference code (<r^2> = 6.424247).
`--kinds`: the same check for all four densities of Distributions.cuh:7-28 (each 0 for r < 0) under the bench rule (CROSS, FAST)
and the reference's (RECIP, PRECISE): 2^22 chains x 4096 measured sweeps = 1.7e10 samples per line."""
import json
import os
import sys

import numpy as np
import torch
from scipy import integrate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import api, lib as L  # noqa: E402


def all_kinds(dev):
    dens = {
        "ws": (L.TARGET_WS, 2.0, 0.54, lambda r, c, a: 1.0 / (1.0 + np.exp((r - c) / a)), 3.0),
        "r2ws": (L.TARGET_R2WS, 2.0, 0.54, lambda r, c, a: r * r / (1.0 + np.exp((r - c) / a)), 2.48),
        "gauss": (L.TARGET_GAUSS, 1.0, 0.0, lambda r, v, _: np.exp(-r * r / (2 * v)), 2.0),
        "r2gauss": (L.TARGET_R2GAUSS, 1.0, 0.0, lambda r, v, _: r * r * np.exp(-r * r / (2 * v)), 1.6),
    }
    blocks, per = 32, 1 << 17
    for kname, (kind, p0, p1, f, width) in dens.items():
        hi = p0 + 60 * p1 if p1 > 0 else 14.0 * np.sqrt(p0)
        z = integrate.quad(lambda r: f(r, p0, p1), 0, hi, epsabs=0, epsrel=1e-13)[0]
        m1 = integrate.quad(lambda r: r * f(r, p0, p1), 0, hi, epsabs=0, epsrel=1e-13)[0] / z
        m2 = integrate.quad(lambda r: r * r * f(r, p0, p1), 0, hi, epsabs=0, epsrel=1e-13)[0] / z
        tg = api.make_target(kind, p0, p1)
        for name, rule, math in (("cross_fast", L.ACCEPT_CROSS, L.MATH_FAST), ("recip_precise", L.ACCEPT_RECIP, L.MATH_PRECISE)):
            r1, r2, acc = [], [], []
            for k in range(blocks):
                eng = api.ChainEngine(per, 1, tg, width, precision=32, accept_rule=rule, math_mode=math, n_lags=0, seed=20261112,
                                      x0=1.0, chain_offset=k * per, center=m1, device=dev)
                eng.run(512)
                eng.run(4096, window_start=True)
                torch.cuda.synchronize()
                q = eng.summary()[0]
                r1.append(q["mean"])
                r2.append(q["variance"] + q["mean"] ** 2)
                acc.append(q["acceptance"])
            r1, r2 = np.array(r1), np.array(r2)
            e1, e2 = r1.std(ddof=1) / np.sqrt(blocks), r2.std(ddof=1) / np.sqrt(blocks)
            print(json.dumps({"target": kname, "p0": p0, "p1": p1, "width": width, "rule": name, "samples": blocks * per * 4096,
                              "acceptance": float(np.mean(acc)), "mean_r": r1.mean(), "mean_r_quadrature": m1, "mean_r_error": e1,
                              "mean_r_sigmas": (r1.mean() - m1) / e1, "mean_r2": r2.mean(), "mean_r2_quadrature": m2,
                              "mean_r2_error": e2, "mean_r2_sigmas": (r2.mean() - m2) / e2}), flush=True)


def main():
    dev = torch.device("cuda:0")
    if "--kinds" in sys.argv:
        return all_kinds(dev)
    c, a = 2.0, 0.54
    w = lambda r, k: r ** (2 + k) / (1.0 + np.exp((r - c) / a))  # noqa: E731
    z = integrate.quad(w, 0, 40, args=(0,), epsabs=0, epsrel=1e-13)[0]
    m1 = integrate.quad(w, 0, 40, args=(1,), epsabs=0, epsrel=1e-13)[0] / z
    m2 = integrate.quad(w, 0, 40, args=(2,), epsabs=0, epsrel=1e-13)[0] / z
    tg = api.make_target(L.TARGET_R2WS, c, a)
    blocks, per = 64, 1 << 18
    for name, rule, math in (("cross_fast", L.ACCEPT_CROSS, L.MATH_FAST), ("recip_fast", L.ACCEPT_RECIP, L.MATH_FAST),
                             ("log_fast", L.ACCEPT_LOG, L.MATH_FAST), ("recip_precise", L.ACCEPT_RECIP, L.MATH_PRECISE)):
        r1, r2 = [], []
        for k in range(blocks):
            eng = api.ChainEngine(per, 3, tg, 2.48, precision=32, accept_rule=rule, math_mode=math, n_lags=0, seed=20261111,
                                  x0=1.0, chain_offset=k * per, center=2.3, device=dev)
            eng.run(512)
            eng.run(4096, window_start=True)
            torch.cuda.synchronize()
            s = eng.summary()
            mean = np.mean([q["mean"] for q in s])
            var = np.mean([q["variance"] for q in s])
            r1.append(mean)
            r2.append(var + mean * mean)   # per coordinate var + mean^2, averaged: exact to O(1e-9) here
        r1, r2 = np.array(r1), np.array(r2)
        e1, e2 = r1.std(ddof=1) / np.sqrt(blocks), r2.std(ddof=1) / np.sqrt(blocks)
        print(json.dumps({"rule": name, "samples": blocks * per * 3 * 4096, "mean_r": r1.mean(), "mean_r_quadrature": m1,
                          "mean_r_error": e1, "mean_r_sigmas": (r1.mean() - m1) / e1, "mean_r2": r2.mean(),
                          "mean_r2_quadrature": m2, "mean_r2_error": e2, "mean_r2_sigmas": (r2.mean() - m2) / e2}))


if __name__ == "__main__":
    main()
