#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run in the build container,
where /root/reference exists; the fixtures, not the reference, travel).

Sources of the expected values:
  ref_*      : the REFERENCE's own code -- stepsize/include/{metropolis,statistics}.hpp
               compiled unmodified into oracle/_ref (oracle/ref_stepsize_shim.cpp), and
               xsection/tools/XSectionAnalysis.py imported from /root/reference.
  oracle_*   : the CPU restatement (oracle/mcmc_oracle.c) for the paths whose reference
               sources need the CUDA toolkit (xsection/include/*.cuh) -- regression
               vectors, "parity unpinned" against reference outputs.
Inputs are seeded numpy draws, stored with the outputs.
"""
import importlib.util
import os
import sys

import numpy as np

sys.dont_write_bytecode = True   # importing XSectionAnalysis.py must not leave a __pycache__ in the read-only reference tree

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

REF_ROOT = "/root/reference"


def main():
    orc = O.Oracle()
    ref = O.Reference()
    rng = np.random.default_rng(20261110)

    # (1) a1 trajectories from the reference: MetropolisHastingsAlgorithm, target PDF() = r^2 WS(2.0, 0.54)
    n = 4096
    normals = rng.standard_normal(n)
    uniforms = rng.random(n)
    out = {"normals": normals, "uniforms": uniforms}
    for step in (0.5, 2.5, 5.0):
        ratio, samples, used = ref.mh(n, step, normals, uniforms)
        tag = str(step).replace(".", "p")
        out[f"ref_samples_{tag}"] = samples
        out[f"ref_acceptance_{tag}"] = np.array([ratio])
        out[f"ref_uniforms_used_{tag}"] = np.array([used])
        # the reference's statistics of its own trajectory (statistics.hpp via main.cpp:27-30)
        out[f"ref_average_{tag}"] = np.array([ref.average(samples)])
        out[f"ref_variance_{tag}"] = np.array([ref.variance(samples)])
        out[f"ref_autocorrelation_{tag}"] = np.array([ref.autocorrelation(samples, k) for k in range(1, 50)])
        # r^2 Gauss(1) (metropolis.hpp:14-17; BASELINE config 0) -- the reference's PDF() is hard-wired to WS,
        # so this trajectory comes from the oracle
        acc, s, bits, used = orc.mh_chain_f64(n, step, O.R2GAUSS, 1.0, 0.0, 1.0, normals, uniforms)
        out[f"oracle_r2gauss_samples_{tag}"] = s
        out[f"oracle_r2gauss_accepted_{tag}"] = np.array([acc])
    r = np.linspace(-1.0, 12.0, 131)
    out["pdf_r"] = r
    out["ref_R2WoodsSaxon_2_0p54"] = np.array([ref.r2ws(x, 2.0, 0.54) for x in r])
    out["ref_R2WoodsSaxon_6p62_0p546"] = np.array([ref.r2ws(x, 6.62, 0.546) for x in r])
    out["ref_R2Gaussian_1"] = np.array([ref.r2gauss(x, 1.0) for x in r])
    np.savez_compressed(os.path.join(HERE, "stepsize_a1.npz"), **out)

    # (2) a4 statistics from the reference on a fixed AR(1) series
    m = 10000
    e = rng.standard_normal(m)
    x = np.empty(m)
    x[0] = 2.3
    for i in range(1, m):
        x[i] = 2.3 + 0.8 * (x[i - 1] - 2.3) + 0.6 * e[i]
    st = {"series": x, "ref_average": np.array([ref.average(x)]), "ref_variance": np.array([ref.variance(x)]),
          "ref_autocorrelation": np.array([ref.autocorrelation(x, k) for k in range(1, 50)])}
    x32 = x.astype(np.float32)
    avg32 = orc.average_f32(x32)
    st["oracle_f32_average"] = np.array([avg32], dtype=np.float32)
    st["oracle_f32_variance_pop"] = np.array([orc.variance_f32(x32, avg32)], dtype=np.float32)
    st["oracle_f32_autocorrelation"] = np.array([orc.autocorrelation_f32(x32, avg32, k) for k in range(1, 50)],
                                                dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "statistics_a4.npz"), **st)

    # (3) sigma integration from the reference's Python (XSectionAnalysis.py:3-28) on a synthetic S_AB table
    spec = importlib.util.spec_from_file_location("XSectionAnalysis", os.path.join(REF_ROOT, "xsection/tools/XSectionAnalysis.py"))
    xsa = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(xsa)
    b = 0.05 + 0.1 * np.arange(200)          # MacroGenerator.ipynb cell 1: b0 = 0.05, db = 0.1, ndbs = 200
    S = 1.0 / (1.0 + np.exp(-(b - 5.2) / 0.45)) * (1 - 0.02 * np.exp(-b))
    Serr = 1e-4 * (1 + np.sin(b) ** 2) * S * (1 - S) + 1e-7
    tot, el, rxn = xsa.SigmaTot(b, S, Serr), xsa.SigmaEl(b, S, Serr), xsa.SigmaRxn(b, S, Serr)
    np.savez_compressed(os.path.join(HERE, "sigma_f1.npz"), b=b, S_AB=S, S_AB_stderr=Serr,
                        ref_sigma=np.array([tot[0], tot[1], el[0], el[1], rxn[0], rxn[1]]))

    # (4) FP32 xsection vectors from the oracle (a5/a7/a10/a12)
    xs = {}
    for nN in (1, 3, 8):
        iters = 96
        draws = np.empty((iters, nN, 2), dtype=np.float32)
        draws[:, :, 0] = rng.standard_normal((iters, nN))
        draws[:, :, 1] = 1.0 - rng.random((iters, nN))
        tg = [(O.R2WS, 2.0, 0.54)] * nN
        xf, bits, trace, margin = orc.warmup_thread_trace(tg, [1.0] * nN, 2.2, iters, draws)
        x2, r2 = orc.integrate_r2_thread(tg, [1.0] * nN, 2.2, iters, draws)
        xs[f"draws_nN{nN}"] = draws
        xs[f"oracle_trace_nN{nN}"] = trace
        xs[f"oracle_bits_nN{nN}"] = bits
        xs[f"oracle_margin_nN{nN}"] = np.array([margin], dtype=np.float32)
        xs[f"oracle_r2mean_nN{nN}"] = r2
    nA = nB = 2
    nN, samples = 4, 64
    d = np.empty((samples, 2, nN, 2), dtype=np.float32)
    d[:, 0, :, 0] = rng.standard_normal((samples, nN))
    d[:, 0, :, 1] = 1.0 - rng.random((samples, nN))
    d[:, 1] = 1.0 - rng.random((samples, nN, 2))
    xs["sab_draws"] = d
    tg = [(O.R2WS, 2.0, 0.54)] * nN
    for bb in (0.0, 1.5, 4.0):
        xf, mean, hits, ylast = orc.integrate_sab_thread(nA, nB, tg, [1.0] * nN, 2.2, samples, bb, d, dmax2=4.3 / np.pi)
        tag = str(bb).replace(".", "p")
        xs[f"oracle_sab_mean_b{tag}"] = np.array([mean], dtype=np.float32)
        xs[f"oracle_sab_hits_b{tag}"] = hits
        xs[f"oracle_sab_state_b{tag}"] = xf
    np.savez_compressed(os.path.join(HERE, "xsection_f32.npz"), **xs)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
