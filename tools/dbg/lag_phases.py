#!/usr/bin/env python3
"""Development: per-workgroup cycles of the lag kernel's prologue / epilogue (MCMC_LAG_PHASES build of the stamped kernel)."""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
dump = "/tmp/lag_stamps.bin"
os.environ["MCMC_STAMP_DUMP"] = dump
from mcmc_amd import api, lib as L  # noqa: E402
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chains = 1 << 22
e = api.ChainEngine(chains, 3, api.make_target(L.TARGET_R2WS, 2.0, 0.54), 2.48, precision=32, accept_rule=L.ACCEPT_CROSS,
                    math_mode=L.MATH_FAST, n_lags=49, center=2.3, device=torch.device("cuda:0"))
e.run(256, n_lags=49)
torch.cuda.synchronize()
api.chain_clock_probe(e, sweeps, n_lags=49)
h = np.fromfile(dump, dtype=np.uint64).reshape(-1, 2)
h = h[h[:, 1] > 0]
pro = (h[:, 0] & np.uint64(0xffffffff)).astype(np.float64)
epi = (h[:, 0] >> np.uint64(32)).astype(np.float64)
tot = h[:, 1].astype(np.float64)
tiles = chains / 256 / 2048
print(json.dumps({"sweeps": sweeps, "workgroups": int(len(h)), "tiles_per_wg": tiles, "total_cycles_per_wg": tot.mean(),
                  "prologue_cycles_per_tile": pro.mean() / tiles, "epilogue_cycles_per_tile": epi.mean() / tiles,
                  "loop_cycles_per_step": (tot.mean() - pro.mean() - epi.mean()) / tiles / sweeps,
                  "sum_wg_cycles_over_768_slots": tot.sum() / 768}))
