import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mcmc_amd import api, lib as L
dev = torch.device("cuda:0")
def run(n_chains, off, t0, n, opt, rule=L.ACCEPT_CROSS, lags=49):
    L.lib.mcmc_set_option(b"MCMC_LAG_MFMA", opt)
    e = api.ChainEngine(n_chains, 2, api.make_target(1, 2.0, 0.54), 2.48, precision=32, accept_rule=rule, math_mode=L.MATH_FAST,
                        n_lags=lags, seed=12345, chain_offset=off, center=2.33, device=dev)
    e.steps_done = t0
    bits = torch.zeros((n, 2, n_chains), dtype=torch.uint8, device=dev)
    st = e.run(n, accept_bits=bits).cpu().numpy().copy()
    torch.cuda.synchronize()
    return st, e.state.cpu().numpy().copy(), bits.cpu().numpy().sum(axis=(0, 2))
for (t0, n) in ((0, 400), (50, 400), (48, 400), (50, 64)):
    for opt in (b"1", b"0"):
        a = run(1000, 0, t0, n, opt)
        h0 = run(500, 0, t0, n, opt); h1 = run(500, 500, t0, n, opt)
        print(t0, n, opt, "acc whole", a[0][:, 2], "bits", a[2], "parts", h0[0][:, 2] + h1[0][:, 2], "bits", h0[2] + h1[2],
              "state eq", np.array_equal(np.concatenate([h0[1], h1[1]], axis=1), a[1]))
# no bits (STORE 0)
for opt in (b"1", b"0"):
    L.lib.mcmc_set_option(b"MCMC_LAG_MFMA", opt)
    res = []
    for (nc, off) in ((1000, 0), (500, 0), (500, 500)):
        e = api.ChainEngine(nc, 2, api.make_target(1, 2.0, 0.54), 2.48, precision=32, accept_rule=L.ACCEPT_CROSS, math_mode=L.MATH_FAST,
                            n_lags=49, seed=12345, chain_offset=off, center=2.33, device=dev)
        e.steps_done = 50
        res.append(e.run(400).cpu().numpy().copy()[:, 2])
    print("store0", opt, res[0], res[1] + res[2])
