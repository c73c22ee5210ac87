#!/usr/bin/env python3
"""Pb-208 + Pb-208: the lane tiling (64 lanes per event as a GA x GB grid, MCMC_SAB_PLAN="64,log2 GA") against the cost
model's choice, for the product + counters kernel at b = 0 / 7 / 13 fm, the product-only kernel and the inelastic flag.
One process: the plan is switched with mcmc_set_option.  One JSON line per (kernel, b)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import api, lib as L  # noqa: E402


def timed(fn, reps=5):
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
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    pb = (L.TARGET_R2WS, 6.62, 0.546)
    T, ns = 1 << 14, 16
    xs = api.XsEngine(208, 208, T, [pb] * 416, accept_rule=L.ACCEPT_RECIP, math_mode=L.MATH_FAST, sigmaNN=6.76, device=dev)
    xs.InitSampleFromPdf()
    xs.WarmupMetropolis(64, stepsize=3.9)
    dmax2 = 6.76 / np.pi
    cases = [("S_AB+hits", b, (lambda b=b: xs.MCIntegrate_S_AB(ns, b, stepsize=3.9, dmax2=dmax2))) for b in (0.0, 7.0, 13.0)]
    cases += [("S_AB", b, (lambda b=b: xs.MCIntegrate_S_AB(ns, b, stepsize=3.9))) for b in (0.0, 7.0, 13.0)]
    cases += [("Inel", b, (lambda b=b: xs.MCIntegrate_Inel(ns, b, dmax2, stepsize=3.9))) for b in (14.0,)]
    for name, b, fn in cases:
        res = {}
        for gl in range(0, 7):
            L.check(L.lib.mcmc_set_option(b"MCMC_SAB_PLAN", f"64,{gl}".encode()))
            try:
                res[f"GA={1 << gl}"] = round(timed(fn), 4)
            except L.McmcError as e:
                res[f"GA={1 << gl}"] = str(e)[:60]
        L.check(L.lib.mcmc_set_option(b"MCMC_SAB_PLAN", None))
        res["model"] = round(timed(fn), 4)
        print(json.dumps({"kernel": name, "b": b, "events": T, "samples": ns, "ms_by_plan": res}), flush=True)


if __name__ == "__main__":
    main()
