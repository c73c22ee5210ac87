#!/usr/bin/env python3
"""Sweep the S_AB lane tiling (G, GA) for a few systems via MCMC_SAB_PLAN; prints ms per launch."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
from mcmc_amd import api, lib as L
nA, T, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tg = (L.TARGET_R2WS, 2.0 if nA < 50 else 6.62, 0.54)
xs = api.XsEngine(nA, nA, T, [tg] * (2 * nA), accept_rule=L.ACCEPT_RECIP, math_mode=L.MATH_FAST, device="cuda:0")
xs.WarmupMetropolis(64, stepsize=2.2)
xs.MCIntegrate_S_AB(S, 1.0, stepsize=2.2); torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); xs.MCIntegrate_S_AB(S, 1.0, stepsize=2.2); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(sorted(ts)[2])
''' % ROOT
for nA, T, S in ((4, 1 << 20, 256), (12, 1 << 18, 64), (40, 1 << 16, 32), (208, 1 << 14, 16)):
    res = {}
    for G in (4, 16, 64):
        for gl in range(0, G.bit_length()):
            env = dict(os.environ, MCMC_SAB_PLAN=f"{G},{gl}")
            r = subprocess.run([sys.executable, "-c", code, str(nA), str(T), str(S)], env=env, capture_output=True, text=True)
            try:
                res[f"{G},{gl}"] = round(float(r.stdout.strip().splitlines()[-1]), 3)
            except Exception:
                res[f"{G},{gl}"] = None
    env = {k: v for k, v in os.environ.items() if k != "MCMC_SAB_PLAN"}
    r = subprocess.run([sys.executable, "-c", code, str(nA), str(T), str(S)], env=env, capture_output=True, text=True)
    print(json.dumps({"nA=nB": nA, "events": T, "samples": S, "ms_by_plan": res, "model_pick_ms": r.stdout.strip()}), flush=True)
