#!/usr/bin/env python3
"""bench.py -- MH proposals/sec of the many-chain Metropolis-Hastings kernel.

One "step" = one launch of the hot path over one batch: `--sweeps` MH sweeps of
every chain (2^24 chains x 3 independent r^2 Woods-Saxon(2.0, 0.54) coordinates
per GPU, the 3-D target of BASELINE.json configs[1] at the chain count the
metric is quoted on), with the streaming sufficient statistics (sum, sum of
squares, accepts) reduced on the device; for N > 1 followed by ONE RCCL
all-reduce of the statistics vector (the path's only exchange step).  Chains are
sharded by global chain id (Philox counter), so results do not depend on N.

Prints ONE JSON line (see the driver contract in the task statement) with
`roofline` (FP32 VALU bound: SURVEY.md 8d) and `cpu_baseline` (the oracle's
restatement of the reference step logic timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PROPOSAL = 24.0      # algorithmic FP32 flop per 1-D r^2 WS proposal
VALU_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: peak FP32 vector


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=1 << 24, help="chains per GPU")
    ap.add_argument("--coords", type=int, default=3)
    ap.add_argument("--sweeps", type=int, default=256, help="MH sweeps per step (launch)")
    ap.add_argument("--step-size", type=float, default=2.48,
                    help="proposal width (2.48 = max-efficiency width of stepsize/results/r2WS_2.0_0.54.png)")
    ap.add_argument("--rule", default="cross", choices=["recip", "nodiv", "log", "cross"],
                    help="accept rule: cross = p' >= U p cross-multiplied (default), recip = the reference's cached-reciprocal form")
    ap.add_argument("--math", default="fast", choices=["fast", "precise"])
    ap.add_argument("--lags", type=int, default=0, help="lag products inside the timed kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-ess", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other kernels of the path")
    return ap.parse_args()


def secondary(dev):
    """Short runs of the other kernels of the path (outside the timed region; rank 0, N = 1 only): reported beside the
    headline, never part of `value`.  tools/bench_xs.py, bench_mvn.py, bench_store.py are the full versions."""
    import numpy as np
    from mcmc_amd import api, lib as L

    def timed(fn, reps=3):
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
        return sorted(ts)[len(ts) // 2] * 1e-3

    out = {}
    try:
        pb = (L.TARGET_R2WS, 6.62, 0.546)
        T, ns = 1 << 14, 16
        xs = api.XsEngine(208, 208, T, [pb] * 416, accept_rule=L.ACCEPT_CROSS, math_mode=L.MATH_FAST, sigmaNN=6.76, device=dev)
        xs.InitSampleFromPdf()
        xs.WarmupMetropolis(64, stepsize=3.9)
        t = timed(lambda: xs.MCIntegrate_S_AB(ns, 13.0, stepsize=3.9))
        out["sab_pbpb_b13_event_samples_per_s"] = T * ns / t
        # nominal A x B pairs of the product; the kernel skips those that provably contribute the factor 1.0f
        out["sab_pbpb_b13_nominal_pair_terms_per_s"] = T * ns * 208 * 208 / t
        t = timed(lambda: xs.MCIntegrate_Inel(ns, 7.0, 6.76 / np.pi, stepsize=3.9))
        out["inel_pbpb_b7_event_samples_per_s"] = T * ns / t
        t = timed(lambda: xs.WarmupMetropolis(256, stepsize=3.9))
        out["nucleon_chain_proposals_per_s"] = T * 416 * 256 / t
        del xs
        n, D, K = 1 << 21, 64, 32
        qd, qo = api.ar1_precision(D, 0.9)
        mv = api.MvnEngine(n, qd, qo, 0.8, 1000, math_mode=L.MATH_FAST, x0=0.0, device=dev)
        obs = torch.empty((K, n), dtype=torch.float32, device=dev)
        t = timed(lambda: mv.run(K, window=(0, 1 << 40), samples_obs=obs))
        out["stress64_lag1000_proposals_per_s"] = n * D * K / t
        out["stress64_lag1000_algorithmic_GBps"] = n * (K * 12 + 2 * 4 * D) / t / 1e9
        del mv, obs
        nc, C, S = 1 << 24, 3, 32
        eng = api.ChainEngine(nc, C, api.make_target(L.TARGET_R2WS, 2.0, 0.54), 2.48, precision=32, accept_rule=L.ACCEPT_CROSS,
                              math_mode=L.MATH_FAST, center=2.3, device=dev)
        smp = torch.empty((S, C, nc), dtype=torch.float32, device=dev)
        t = timed(lambda: eng.run(S, samples=smp))
        out["chain_writeback_proposals_per_s"] = nc * C * S / t
        out["chain_writeback_algorithmic_GBps"] = (nc * C * S * 4 + 2 * 4 * nc * C) / t / 1e9
    except Exception as e:  # the headline line must not depend on these
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def cpu_baseline(args, seconds):
    """The oracle's FP32 restatement of Metropolis.cuh:40-52 (+ the library's Philox draws and the
    same three accumulators), one thread per core share; bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    orc = O.Oracle()
    cores = min(16, os.cpu_count() or 1)
    sweeps = args.sweeps
    t0 = time.perf_counter()
    orc.baseline_chains_f32(0, 64, args.coords, O.R2WS, 2.0, 0.54, args.step_size, 314159265, 0, sweeps)
    probe = time.perf_counter() - t0
    per_chain = probe / 64
    chains_per_thread = max(16, int(seconds / per_chain))

    def work(i):
        return orc.baseline_chains_f32(i * chains_per_thread, chains_per_thread, args.coords, O.R2WS, 2.0, 0.54,
                                       args.step_size, 314159265, 0, sweeps)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    proposals = cores * chains_per_thread * args.coords * sweeps
    return {"value": proposals / dt, "unit": "proposals/s", "cores": cores, "kind": "port",
            "sample": f"{cores * chains_per_thread} chains x {args.coords} coords x {sweeps} sweeps "
                      f"(oracle/mcmc_oracle.c orc_baseline_chains_f32, one thread per core, {dt:.1f} s)"}


def cpu_reference_f64(step_size):
    """The reference's OWN MetropolisHastingsAlgorithm (stepsize/include/metropolis.hpp:28-50, built unmodified into
    oracle/_ref by oracle/Makefile), one chain on one core, FP64.  Its RNG (clock-seeded mt19937, random.hpp) is
    replaced by pre-drawn arrays in that build, so the figure EXCLUDES random-number generation."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    if not O.have_ref():
        return None
    ref = O.Reference()
    n = 2_000_000
    rng = np.random.default_rng(1)
    z, u = rng.standard_normal(n), rng.random(n)
    ref.mh(1000, step_size, z, u)
    t0 = time.perf_counter()
    ref.mh(n, step_size, z, u)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "proposals/s", "cores": 1, "kind": "reference",
            "sample": f"{n} iterations of the reference's MetropolisHastingsAlgorithm (FP64, one chain, injected draws: "
                      f"RNG cost excluded), {dt:.2f} s"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # one rank per GPU; the modulo only matters when more ranks than GPUs are rehearsed on one card
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("MCMC_BENCH_BACKEND", "nccl")   # "nccl" is RCCL; "gloo" for a one-GPU rehearsal
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    trace_on = os.environ.get("MCMC_BENCH_TRACE") == "1"

    def trace(msg):
        if trace_on:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    trace("process group up")
    from mcmc_amd import api, dist as D, lib as L
    rule = {"recip": L.ACCEPT_RECIP, "nodiv": L.ACCEPT_NODIV, "log": L.ACCEPT_LOG, "cross": L.ACCEPT_CROSS}[args.rule]
    math = L.MATH_FAST if args.math == "fast" else L.MATH_PRECISE
    target = api.make_target(L.TARGET_R2WS, 2.0, 0.54)   # PDF(), metropolis.hpp:19-23 / main.cu:14-19
    eng = api.ChainEngine(args.chains, args.coords, target, args.step_size, precision=32, accept_rule=rule,
                          math_mode=math, n_lags=args.lags, seed=314159265, x0=1.0,
                          chain_offset=rank * args.chains, center=2.3, device=dev)

    def step():
        st = eng.run(args.sweeps, window_start=True)
        D.allreduce_stats(st)          # the one collective per batch: sufficient statistics only (no-op for N = 1)
        return st

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    trace("engine created")
    for _ in range(args.warmup):
        step()
    trace("warm-up steps queued")
    fence()
    trace("fence passed")
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record()
        st = eng.run(args.sweeps, window_start=True)
        evs[i][1].record()
        D.allreduce_stats(st)
    fence()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, device=dev)
    trace(f"timed region done: {elapsed:.3f} s")
    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / args.steps

    proposals_per_step = world * args.chains * args.coords * args.sweeps
    value = proposals_per_step * args.steps / elapsed
    summ = eng.summary(stats_host=st.cpu().numpy(), n_chains=world * args.chains, n_steps=args.sweeps)

    out = None
    if rank == 0:
        launch_props = args.chains * args.coords * args.sweeps
        # HBM bytes per launch from the committed PMC passes of this same workload (tools/gpu_profile.sh;
        # FETCH_SIZE / WRITE_SIZE in their own runs, FETCH doubled per MI355X_MICROARCH.md), else null
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_chain_kernel_v6.json")))
            if args.chains == 1 << 24 and args.coords == 3 and args.lags == 0:
                traffic = pm["derived"]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        # VALU issue occupancy: issue cycles of the hot loop's instruction mix (tools/isa_mix.py, costs from
        # tools/ubench2.hip) over the SIMD cycles one wave-proposal actually took (1024 SIMDs, clock from the PMC run)
        valu = None
        try:
            mix = json.load(open(os.path.join(ROOT, "profiles", "r1_chain_kernel_isa_mix.json")))
            ghz = pm["derived"]["clock_GHz"]
            if args.rule == "cross" and args.math == "fast" and args.lags == 0:
                cyc = 1024 * ghz * 1e9 * 64 * (kernel_ms * 1e-3) / launch_props
                valu = {"measured_cycles_per_wave_proposal": cyc, "clock_GHz": ghz,
                        "ideal_issue_cycles": mix["issue_cycles_per_wave_proposal_ideal"],
                        "isolated_cost_issue_cycles": mix["issue_cycles_per_wave_proposal_isolated_costs"],
                        "frac_of_ideal_issue": mix["issue_cycles_per_wave_proposal_ideal"] / cyc,
                        "frac_of_isolated_cost_issue": mix["issue_cycles_per_wave_proposal_isolated_costs"] / cyc,
                        "valu_insts_per_proposal": mix["valu_insts_per_proposal"]}
        except (OSError, KeyError, ValueError, NameError):
            pass
        achieved = FLOP_PER_PROPOSAL * launch_props / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "MH proposals/sec", "value": value, "unit": "proposals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"mh_chain r2WoodsSaxon(2.0,0.54) x{args.coords} coords, "
                                   f"{args.chains} chains/GPU, {args.sweeps} sweeps/launch, step {args.step_size}",
                       "chains_per_gpu": args.chains, "coords": args.coords, "sweeps_per_step": args.sweeps,
                       "accept_rule": args.rule, "math": args.math, "lags_in_kernel": args.lags,
                       "rng": "philox4x32-10", "parallelism": f"chains sharded over {world} GPU(s)"},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / VALU_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": "mcmc::chain_kernel", "kernel_ms": kernel_ms,
                         "flop_per_proposal": FLOP_PER_PROPOSAL, "proposals_per_launch": launch_props,
                         "valu_issue": valu,
                         "note": "FP32 VALU-issue-bound (state in registers, no per-step HBM traffic). achieved = 24 "
                                 "algorithmic FP32 flop per proposal (SURVEY 8d) / kernel time; Philox integer work and "
                                 "quarter-rate transcendentals are not counted as flops, so frac understates VALU "
                                 "occupancy: valu_issue sets the issue cycles of the loop's instruction mix against the "
                                 "cycles measured (the '% of VALU peak' of SURVEY 8d; DESIGN.md 4.1). traffic = HBM bytes per launch from the PMC passes under profiles/"},
            "estimates": {"mean_r": summ[0]["mean"], "var_r": summ[0]["variance"],
                          "acceptance": summ[0]["acceptance"]},
        }
    # ESS/sec: tau_int = 1 + 2 sum rho_k from the streaming lag-49 accumulators at the same width, on a chain
    # subset after warm-up (outside the timed region); the same launches give the throughput of the kernel
    # with all 49 lag products carried by every chain.
    if not args.no_ess:
        sub = min(args.chains, 1 << 22)
        e2 = api.ChainEngine(sub, args.coords, target, args.step_size, precision=32, accept_rule=rule, math_mode=math,
                             n_lags=L.MAX_LAGS, seed=314159265, x0=1.0, chain_offset=rank * args.chains, center=2.3,
                             device=dev)
        e2.run(512, n_lags=0)
        e2.run(256, window_start=True, n_lags=L.MAX_LAGS)
        torch.cuda.synchronize()
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        e2.run(1024, window_start=True, n_lags=L.MAX_LAGS)
        eb.record()
        torch.cuda.synchronize()
        tau = e2.summary()[0]["tau_int"]
        if out is not None:
            out["ess_per_sec"] = value / tau
            out["tau_int"] = tau
            out["lag49_proposals_per_sec_per_gpu"] = sub * args.coords * 1024 / (ea.elapsed_time(eb) * 1e-3)
    if rank == 0 and world == 1 and not args.no_secondary:
        out["secondary"] = secondary(dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        out["cpu_reference_f64"] = cpu_reference_f64(args.step_size)
    elif rank == 0:
        out["cpu_baseline"] = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
