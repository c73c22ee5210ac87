#!/usr/bin/env python3
"""tools/tune_c3.py -- SURVEY 8d C3 "report tuned step too": the device step-size study for the heavy-ion radial
densities (Pb-208, Au-197 literature Woods-Saxon) against the reference's constant 2.2 (main.cu:149)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import api, lib as L  # noqa: E402

for name, c, a in (("Pb-208", 6.62, 0.546), ("Au-197", 6.38, 0.535), ("main.cu:14-19", 2.0, 0.54)):
    tg = api.make_target(L.TARGET_R2WS, c, a)
    lo, hi = 0.25 * (c + 2 * a), 2.0 * (c + 2 * a)
    best, w, acc, tau = api.TuneStepSize(tg, lo, hi, n_widths=15, n_chains=1 << 16, n_warmup=2000, n_steps=4000)
    _, _, acc22, tau22 = api.TuneStepSize(tg, 2.2, 2.3, n_widths=3, n_chains=1 << 16, n_warmup=2000, n_steps=4000)
    _, _, accb, taub = api.TuneStepSize(tg, best, best * 1.01, n_widths=3, n_chains=1 << 16, n_warmup=2000, n_steps=4000)
    print(json.dumps({"density": f"r2WS({c},{a}) {name}", "best_width": best, "tau_int_best": taub[0], "acceptance_best": accb[0],
                      "tau_int_2.2": tau22[0], "acceptance_2.2": acc22[0], "samples_saved": 1 - taub[0] / tau22[0],
                      "widths": w.round(3).tolist(), "tau_int": tau.round(3).tolist(), "acceptance": acc.round(4).tolist(),
                      "note": "tau_int = 1 + 2 sum_{k<=49} rho_k (truncated at the reference's 49 lags)"}))
