"""The optional Philox4x32-7 build of the library (`make philox7` -> mcmc_amd/libmcmc_hip_philox7.so): its draws equal
the oracle's restatement with 7 rounds (itself pinned by Random123's 7-round known-answer vectors), and its chains
sample the same density.  Runs in a child process: the round count is compiled into the library, and this process has
the default (10-round) build loaded."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB7 = os.path.join(ROOT, "mcmc_amd", "libmcmc_hip_philox7.so")

CHILD = r'''
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["MCMC_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MCMC_ROOT"], "tests"))
import oracle_lib as O
from mcmc_amd import api, lib as L
assert L.lib.mcmc_philox_rounds() == 7 and b"philox4x32-7" in L.lib.mcmc_version()
orc = O.Oracle(); orc.set_philox_rounds(7)
dev = torch.device("cuda:0")
n_chains, n_steps, coord, seed, off = 64, 45, 2, 314159265, 5
bufs = [torch.zeros((n_steps, n_chains), dtype=torch.float32, device=dev) for _ in range(6)]
L.check(L.lib.mcmc_launch_draws(seed, 1000, n_chains, coord, off, n_steps, L.MATH_PRECISE, *[C.c_void_p(t.data_ptr()) for t in bufs], None))
torch.cuda.synchronize()
z, u, a0, a1, sz, su = [t.cpu().numpy() for t in bufs]
bad = 0
for c in (0, 5, 63):
    for i in range(n_steps):
        zn, un = orc.mh_draws_f32(seed, 1000 + c, coord, off + i)
        b0, b1 = orc.angle_draws_f32(seed, 1000 + c, coord, off + i)
        szn, sun = orc.sab_mh_draws_f32(seed, 1000 + c, coord, off + i)
        bad += not (u[i, c] == un and a0[i, c] == b0 and a1[i, c] == b1 and su[i, c] == sun)
        bad += abs(z[i, c] - zn) > 2e-6 * max(1.0, abs(zn)) or abs(sz[i, c] - szn) > 2e-6 * max(1.0, abs(szn))
orc.set_philox_rounds(10)
zn10, un10 = orc.mh_draws_f32(seed, 1000, coord, off)
differs = bool(un10 != u[0, 0])
eng = api.ChainEngine(1 << 20, 3, api.make_target(L.TARGET_R2WS, 2.0, 0.54), 2.2, precision=32, accept_rule=L.ACCEPT_CROSS,
                      math_mode=L.MATH_FAST, device=dev)
eng.run(300); eng.run(400); torch.cuda.synchronize()
ex2 = [s["variance"] + s["mean"] ** 2 for s in eng.summary()]
print(json.dumps({"bad": int(bad), "differs_from_10_rounds": differs, "ex2": ex2}))
'''


def test_philox7_build_draws_and_known_answer():
    if not os.path.exists(LIB7):
        pytest.skip("mcmc_amd/libmcmc_hip_philox7.so not built (make philox7)")
    env = {**os.environ, "MCMC_HIP_LIB": LIB7, "MCMC_ROOT": ROOT}
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["bad"] == 0 and d["differs_from_10_rounds"]
    for v in d["ex2"]:
        assert abs(v - 6.424247) < 3.5e-3, v      # <r^2> of r^2 WS(2.0, 0.54), as test_known_answer_r2ws_large
