#!/usr/bin/env python3
"""tools/sab_analytic.py -- the S_AB kernels against ANALYTIC profile averages at high statistics (the formulas and their
derivation: tests/test_sab_known_answer.py).  R2Gaussian densities: one nucleon on 1 / 3 / 200; for one on 200 also the
black-disk inelastic fraction and the mean number of colliding pairs.  One JSON line per (system, b)."""
import json
import os
import sys

import numpy as np
import torch
from scipy import integrate, stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mcmc_amd import api, lib as L  # noqa: E402
from test_sab_known_answer import BETA, SIGMA, s_one_on_n, s_one_on_one  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    dmax2 = SIGMA / np.pi
    for n, v, T, samples, reps, bs in ((1, 1.3, 1 << 18, 256, 4, (0.0, 0.6, 1.2, 2.5)), (3, 0.8, 1 << 18, 256, 4, (0.0, 0.6, 1.2, 2.5)),
                                       (200, 25.0, 1 << 14, 128, 4, (0.0, 4.0, 9.0, 14.0))):
        for mode, rule, math in (("cross_fast", L.ACCEPT_CROSS, L.MATH_FAST), ("recip_precise", L.ACCEPT_RECIP, L.MATH_PRECISE)):
            xs = api.XsEngine(1, n, T, [(L.TARGET_R2GAUSS, v, 0.0)] * (1 + n), stepsize=1.8 * np.sqrt(v), accept_rule=rule,
                              math_mode=math, sigmaNN=SIGMA, beta=BETA, seed=4242, device=dev)
            xs.InitSampleFromPdf()
            xs.WarmupMetropolis(256)
            for b in bs:
                acc, hits = [], np.zeros(4)
                for _ in range(reps):
                    res, _, h = xs.MCIntegrate_S_AB(samples, b, dmax2=dmax2 if n == 200 else None)
                    torch.cuda.synchronize()
                    acc.append(res.double().cpu().numpy())
                    if h is not None:
                        hits += h.cpu().numpy()
                r = np.mean(acc, axis=0)                 # per event: the mean over reps x samples of ITS chain
                err = r.std(ddof=1) / np.sqrt(T)
                want = s_one_on_one(b, v) if n == 1 else s_one_on_n(b, v, n)
                out = {"system": f"1 on {n}, R2Gaussian v={v}", "mode": mode, "b": b, "event_samples": T * samples * reps,
                       "S_AB": r.mean(), "S_AB_analytic": want, "stderr": err, "sigmas": (r.mean() - want) / err}
                if n == 200:
                    q = lambda x, y: stats.ncx2.cdf(dmax2 / v, 2, ((b + x) ** 2 + y * y) / v)  # noqa: E731
                    f = lambda y, x: np.exp(-(x * x + y * y) / (2.0 * v)) / (2.0 * np.pi * v) * (1.0 - q(x, y)) ** n  # noqa: E731
                    rr = 8.0 * np.sqrt(v)
                    out.update({"p_inel": hits[1] / hits[3], "p_inel_analytic": 1.0 - integrate.dblquad(f, -rr, rr, -rr, rr, epsabs=1e-9, epsrel=1e-9)[0],
                                "ncoll_per_sample": hits[0] / hits[3], "ncoll_analytic": n * stats.ncx2.cdf(dmax2 / (2 * v), 2, b * b / (2 * v))})
                print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
