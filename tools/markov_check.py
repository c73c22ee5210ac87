#!/usr/bin/env python3
"""tools/markov_check.py -- the chain kernels' streaming statistics against the transition operator of the random-walk
Metropolis chain (tests/markov_reference.py) at high statistics: 2^22 chains x 3 coordinates x 8192 sweeps = 1.0e11 samples
per line.  Acceptance, variance, the 49 lag-k autocorrelations and tau_int; both lag kernels (matrix cores / VALU)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import markov_reference as M  # noqa: E402
from mcmc_amd import api, lib as L  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n_chains, n_coords, n_steps = 1 << 22, 3, 8192
    for kind, code, p0, p1, width, upper in (("r2ws", L.TARGET_R2WS, 2.0, 0.54, 2.48, 14.0), ("r2ws", L.TARGET_R2WS, 2.0, 0.54, 5.0, 14.0),
                                             ("r2gauss", L.TARGET_R2GAUSS, 1.0, 0.0, 1.6, 9.0), ("gauss", L.TARGET_GAUSS, 1.0, 0.0, 2.0, 9.0)):
        want, left = M.richardson(M.density(kind, p0, p1), width, upper, 2000)
        for mfma in ("1", "0"):
            L.check(L.lib.mcmc_set_option(b"MCMC_LAG_MFMA", mfma.encode()))
            eng = api.ChainEngine(n_chains, n_coords, api.make_target(code, p0, p1), width, precision=32, accept_rule=L.ACCEPT_CROSS,
                                  math_mode=L.MATH_FAST, n_lags=L.MAX_LAGS, seed=777, x0=1.0, center=want["mean"], device=dev)
            eng.run(1024, n_lags=0)
            eng.run(n_steps, window_start=True, n_lags=L.MAX_LAGS)
            torch.cuda.synchronize()
            got = eng.summary()
            L.lib.mcmc_set_option(b"MCMC_LAG_MFMA", None)
            n = n_chains * n_coords * n_steps
            rho = np.mean([g["rho"] for g in got], axis=0)
            se = np.sqrt(want["tau_int"] / n)
            print(json.dumps({
                "target": f"{kind}({p0},{p1})", "width": width, "lag_kernel": "mfma" if mfma == "1" else "valu", "samples": n,
                "operator_discretisation_left": left,
                "acceptance": float(np.mean([g["acceptance"] for g in got])), "acceptance_operator": want["acceptance"],
                "variance": float(np.mean([g["variance"] for g in got])), "variance_operator": want["variance"],
                "tau_int": float(np.mean([g["tau_int"] for g in got])), "tau_int_operator": want["tau_int"],
                "rho_1": float(rho[0]), "rho_1_operator": float(want["rho"][0]),
                "max_abs_rho_deviation": float(np.max(np.abs(rho - want["rho"]))), "rho_standard_error": float(se),
                "max_rho_deviation_in_sigmas": float(np.max(np.abs(rho - want["rho"])) / se)}), flush=True)


if __name__ == "__main__":
    main()
