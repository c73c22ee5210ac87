#!/usr/bin/env python3
"""tools/bench_xs.py -- timings of the xsection kernels on one GPU.

C3: WarmupMetropolis / MCIntegrate_R2 for Au-197 / Pb-208 (radial chains of every nucleon).
C4: MCIntegrate_S_AB Pb-208 + Pb-208 (43 264 pair terms per event sample).
Prints one JSON object per measurement.  Not the headline bench (bench.py); feeds DESIGN.md.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import api, lib as L  # noqa: E402

PB = (L.TARGET_R2WS, 6.62, 0.546)
AU = (L.TARGET_R2WS, 6.38, 0.535)
HE = (L.TARGET_R2WS, 2.0, 0.54)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--events", type=int, default=1 << 14)
    ap.add_argument("--samples", type=int, default=16)
    ap.add_argument("--rule", default="recip")
    ap.add_argument("--which", default="c3,c4,he")
    ap.add_argument("--bs", default="0,7,14", help="impact parameters (fm) of the Pb-Pb S_AB launches")
    ap.add_argument("--hits-b", type=float, default=7.0, help="impact parameter (fm) of the product + counters launches")
    ap.add_argument("--inel-bs", default="0,7,14,18", help="impact parameters (fm) of the Pb-Pb inelastic-only launches")
    args = ap.parse_args()
    rule = {"recip": L.ACCEPT_RECIP, "nodiv": L.ACCEPT_NODIV, "cross": L.ACCEPT_CROSS, "log": L.ACCEPT_LOG}[args.rule]
    dev = torch.device("cuda:0")
    which = args.which.split(",")
    if "c3" in which:
        for name, tg, A in (("Pb-208", PB, 208), ("Au-197", AU, 197)):
            T = (1 << 22) // A
            xs = api.XsEngine(A, 0, T, [tg] * A, accept_rule=rule, math_mode=L.MATH_FAST, device=dev)
            ms = timed(lambda: xs.WarmupMetropolis(256, stepsize=2.2), args.reps)
            print(json.dumps({"kernel": "WarmupMetropolis", "nucleus": name, "events": T, "coord_chains": T * A,
                              "sweeps": 256, "ms": ms, "proposals_per_s": T * A * 256 / (ms * 1e-3)}))
            ms = timed(lambda: xs.MCIntegrate_R2(256, stepsize=2.2), args.reps)
            rm = xs.runMean.cpu().numpy()
            print(json.dumps({"kernel": "MCIntegrate_R2", "nucleus": name, "events": T, "sweeps": 256, "ms": ms,
                              "proposals_per_s": T * A * 256 / (ms * 1e-3), "rms_mean": float((rm.mean()) ** 0.5)}))
    if "c4" in which:
        T = args.events
        xs = api.XsEngine(208, 208, T, [PB] * 416, accept_rule=rule, math_mode=L.MATH_FAST, sigmaNN=6.76, device=dev)
        xs.WarmupMetropolis(512, stepsize=2.2)
        for b in [float(v) for v in args.bs.split(",")]:
            ms = timed(lambda: xs.MCIntegrate_S_AB(args.samples, b, stepsize=2.2), args.reps)
            n = T * args.samples
            print(json.dumps({"kernel": "MCIntegrate_S_AB", "system": "Pb-208+Pb-208", "b": b, "events": T,
                              "samples": args.samples, "ms": ms, "event_samples_per_s": n / (ms * 1e-3),
                              # nominal = all A x B pairs of the product, including those the kernel skips (culled
                              # as exact factors 1.0f, or behind a zero-product exit); how many it evaluates is not counted
                              "nominal_pairs_per_s": n * 208 * 208 / (ms * 1e-3), "evaluated_pairs_per_s": None,
                              "S_AB": float(xs.sabMean.item())}))
        n = T * args.samples
        for cull in ("1", "0"):
            # product + counters at b = 7 fm: bounding boxes cull nothing and the counters forbid the zero-product
            # exit, so with the culling off every nominal pair IS evaluated (evaluated = nominal)
            L.check(L.lib.mcmc_set_option(b"MCMC_SAB_CULL", cull.encode()))
            ms = timed(lambda: xs.MCIntegrate_S_AB(args.samples, args.hits_b, stepsize=2.2, dmax2=6.76 / 3.14159265), args.reps)
            print(json.dumps({"kernel": "MCIntegrate_S_AB+hits", "system": "Pb-208+Pb-208", "b": args.hits_b, "cull": cull == "1",
                              "events": T, "samples": args.samples, "ms": ms, "event_samples_per_s": n / (ms * 1e-3),
                              "nominal_pairs_per_s": n * 208 * 208 / (ms * 1e-3),
                              "evaluated_pairs_per_s": n * 208 * 208 / (ms * 1e-3) if cull == "0" else None,
                              "hits": xs.hits.cpu().tolist(), "S_AB": float(xs.sabMean.item())}))
        L.lib.mcmc_set_option(b"MCMC_SAB_CULL", None)
        for b in [float(v) for v in args.inel_bs.split(",")]:
            ms = timed(lambda: xs.MCIntegrate_Inel(args.samples, b, 6.76 / 3.14159265, stepsize=2.2), args.reps)
            h = xs.hits.cpu().tolist()
            print(json.dumps({"kernel": "MCIntegrate_Inel", "system": "Pb-208+Pb-208", "b": b, "events": T,
                              "samples": args.samples, "ms": ms, "event_samples_per_s": n / (ms * 1e-3),
                              "p_inel": h[0] / h[1]}))
    if "he" in which:
        T = 1 << 20
        xs = api.XsEngine(4, 4, T, [HE] * 8, accept_rule=rule, math_mode=L.MATH_FAST, device=dev)
        xs.WarmupMetropolis(256, stepsize=2.2)
        ms = timed(lambda: xs.MCIntegrate_S_AB(256, 1.0, stepsize=2.2), args.reps)
        n = T * 256
        print(json.dumps({"kernel": "MCIntegrate_S_AB", "system": "He-4+He-4 (main.cu:38-41)", "events": T, "samples": 256,
                          "ms": ms, "event_samples_per_s": n / (ms * 1e-3), "S_AB": float(xs.sabMean.item())}))


if __name__ == "__main__":
    main()
