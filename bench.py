#!/usr/bin/env python3
"""bench.py -- MH proposals/sec of the many-chain Metropolis-Hastings path.

`python bench.py --gpus N --steps K --warmup W [--workload chain|nucleon|stress]`

One "step" = one launch of the hot path over one batch, followed (N > 1) by ONE
RCCL all-reduce of the sufficient statistics (the path's only exchange step).
Work is sharded by GLOBAL chain / event id (Philox counter), `--chains` is PER
GPU (weak scaling), so results do not depend on N.

  chain    (default; BASELINE.json metric, configs[1] target at the metric's chain count)
           2^24 chains x 3 independent r^2 Woods-Saxon(2.0, 0.54) coordinates per GPU, `--sweeps`
           sweeps per launch, streaming sum / sum of squares / accepts reduced on the device.
  nucleon  (configs[2]) radial chains of every nucleon of Pb-208 (or --nucleus au): 2^22 coordinate
           chains per GPU, one MCIntegrate_R2 launch (IntegrateR2.cuh:7-61) + device row means.
  stress   (configs[4]) 2^21 chains per GPU on the 64-D Gaussian Sigma_ij = 0.9^|i-j|, lag-1000 ring in
           HBM, observable written back every sweep.

N ranks: when WORLD_SIZE is unset and --gpus N > 1 this process starts N ranks ITSELF
(`python -m torch.distributed.run ... bench.py ...`, before anything here touches the GPU) and
exits with their code; under an external launcher the world size must equal --gpus or every rank
refuses.  `n_gpus` in the line is the process group's size.

Prints ONE JSON line (driver contract) with `roofline` (FP32 VALU bound: SURVEY.md 8d) and, at
N = 1, `cpu_baseline` (the oracle's restatement of the reference step logic on the host cores).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PROPOSAL = 24.0      # algorithmic FP32 flop per 1-D r^2 WS proposal
FLOP_R2_ACCUMULATE = 6.0      # "+6" for the Kahan r^2 accumulation of MCIntegrate_R2
FLOP_MVN_PROPOSAL = 22.0      # single-coordinate update on the tridiagonal Gaussian
VALU_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: peak FP32 vector
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E
NUCLEI = {"pb": (208, 6.62, 0.546), "au": (197, 6.38, 0.535)}   # literature Woods-Saxon sets


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="chain", choices=["chain", "nucleon", "stress"])
    ap.add_argument("--chains", type=int, default=None,
                    help="chains per GPU (default 2^24 chain, 2^22 coordinate chains nucleon, 2^21 stress)")
    ap.add_argument("--coords", type=int, default=3)
    ap.add_argument("--sweeps", type=int, default=None, help="MH sweeps per step (default 256; stress 64)")
    ap.add_argument("--step-size", type=float, default=None,
                    help="proposal width (chain: 2.48 = max-efficiency width of stepsize/results/r2WS_2.0_0.54.png; "
                         "nucleon: 2.2 = main.cu:149; stress: 0.8)")
    ap.add_argument("--rule", default="cross", choices=["recip", "nodiv", "log", "cross"],
                    help="accept rule: cross = p' >= U p cross-multiplied (default), recip = the reference's cached-reciprocal form")
    ap.add_argument("--math", default="fast", choices=["fast", "precise"])
    ap.add_argument("--lags", type=int, default=0, help="lag products inside the timed kernel (chain)")
    ap.add_argument("--nucleus", default="pb", choices=sorted(NUCLEI), help="nucleon workload: Pb-208 or Au-197")
    ap.add_argument("--full", action="store_true",
                    help="stress workload: write the whole 64-D state back every sweep (samples_full[sweep][coord][chain], the "
                         "HBM-heavy form of configs[4]); the line's roofline is then the HBM one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-ess", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other kernels of the path")
    ap.add_argument("--no-clock", action="store_true", help="skip the in-run shader clock measurement")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.chains is None:
        args.chains = {"chain": 1 << 24, "nucleon": 1 << 22, "stress": 1 << 21}[args.workload]
    if args.full and args.workload != "stress":
        ap.error("--full belongs to --workload stress")
    if args.sweeps is None:
        args.sweeps = (32 if args.full else 64) if args.workload == "stress" else 256
    if args.step_size is None:
        args.step_size = {"chain": 2.48, "nucleon": 2.2, "stress": 0.8}[args.workload]
    return args


# ---------------------------------------------------------------------------------------------------
# N ranks from one command line
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(n_ranks, argv, port):
    """The command that starts `n_ranks` ranks of this file (one per GPU) on this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def launch_ranks(args, argv):
    """Parent of an N-rank run: starts the ranks as children and returns their exit code.  Nothing in this
    process has touched the GPU (not even torch is imported yet)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL on this pool
    env["MCMC_BENCH_LAUNCHED_RANKS"] = str(args.gpus)
    return subprocess.run(launcher_command(args.gpus, argv, free_port()), env=env).returncode


def world_from_env(args):
    """(world, rank, local_rank) of this process; refuses when the launcher's world size is not --gpus."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: world size {world} (WORLD_SIZE) does not equal --gpus {args.gpus}; "
                         f"start it as `python bench.py --gpus N` (it launches its own ranks) or with "
                         f"--nproc-per-node equal to --gpus")
    return world, rank, local_rank


# ---------------------------------------------------------------------------------------------------
def secondary(dev):
    """Short runs of the other kernels of the path (outside the timed region; rank 0, N = 1 only): reported beside the
    headline, never part of `value`.  tools/bench_xs.py, bench_mvn.py, bench_store.py are the full versions."""
    import numpy as np
    import torch
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
        mv.run(1024, window=(0, 1 << 40))   # fill the lag ring, so that the timed launches read it
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
        # the same launch in the pair layout [S / 2][C][nc][2]: one 64-bit store per lane and step pair (A/B of the store path)
        L.check(L.lib.mcmc_set_option(b"MCMC_SAMPLES_PAIRS", b"1"))
        try:
            t = timed(lambda: eng.run(S, samples=smp))
        finally:
            L.lib.mcmc_set_option(b"MCMC_SAMPLES_PAIRS", None)
        out["chain_writeback_pairs_proposals_per_s"] = nc * C * S / t
        out["chain_writeback_pairs_algorithmic_GBps"] = (nc * C * S * 4 + 2 * 4 * nc * C) / t / 1e9
        del eng, smp
        # the headline workload under the log-space rule (north_star (1)) and under the reference's cached-reciprocal rule
        for name, rule in (("log", L.ACCEPT_LOG), ("recip", L.ACCEPT_RECIP)):
            eng = api.ChainEngine(nc, C, api.make_target(L.TARGET_R2WS, 2.0, 0.54), 2.48, precision=32, accept_rule=rule,
                                  math_mode=L.MATH_FAST, center=2.3, device=dev)
            t = timed(lambda: eng.run(256))
            out[f"chain_{name}_rule_proposals_per_s"] = nc * C * 256 / t
            del eng
    except Exception as e:  # the headline line must not depend on these
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def philox7_rate(args):
    """The same workload with the optional Philox4x32-7 build of the library (`make philox7`; the default and every parity
    test is Philox4x32-10): a child process with MCMC_HIP_LIB pointing at it.  None when that library is not built."""
    lib7 = os.path.join(ROOT, "mcmc_amd", "libmcmc_hip_philox7.so")
    if not os.path.exists(lib7) or os.environ.get("MCMC_HIP_LIB"):
        return None
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-ess",
           "--no-secondary", "--no-clock", "--chains", str(args.chains), "--coords", str(args.coords), "--sweeps", str(args.sweeps),
           "--rule", args.rule, "--math", args.math]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env={**os.environ, "MCMC_HIP_LIB": lib7})
        return json.loads(r.stdout.strip().splitlines()[-1])["value"] if r.returncode == 0 else None
    except Exception:  # noqa: BLE001 -- the headline line must not depend on this
        return None


def cpu_cores():
    """Threads for the CPU baseline: the box's CPU share (OMP_NUM_THREADS is set to it there), at most 16."""
    share = os.environ.get("OMP_NUM_THREADS", "")
    n = int(share) if share.isdigit() and int(share) > 0 else (os.cpu_count() or 1)
    return max(1, min(16, n, os.cpu_count() or 1))


def cpu_baseline(args, seconds, target=(2.0, 0.54), note=""):
    """The oracle's FP32 restatement of Metropolis.cuh:40-52 (+ the library's Philox draws and the
    same three accumulators), one thread per core share; bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    orc = O.Oracle()
    cores = cpu_cores()
    sweeps = args.sweeps
    coords = args.coords if args.workload == "chain" else 1
    t0 = time.perf_counter()
    orc.baseline_chains_f32(0, 64, coords, O.R2WS, target[0], target[1], args.step_size, 314159265, 0, sweeps)
    probe = time.perf_counter() - t0
    per_chain = probe / 64
    chains_per_thread = max(16, int(seconds / per_chain))

    def work(i):
        return orc.baseline_chains_f32(i * chains_per_thread, chains_per_thread, coords, O.R2WS, target[0], target[1],
                                       args.step_size, 314159265, 0, sweeps)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    proposals = cores * chains_per_thread * coords * sweeps
    return {"value": proposals / dt, "unit": "proposals/s", "cores": cores, "nproc": os.cpu_count(), "kind": "port",
            "sample": f"{cores * chains_per_thread} chains x {coords} coords x {sweeps} sweeps "
                      f"(oracle/mcmc_oracle.c orc_baseline_chains_f32, one thread per core, {dt:.1f} s){note}"}


def cpu_baseline_mvn(args, seconds):
    """The oracle's restatement of the stress sweep (orc_mvn_chain, Philox draws) on the host cores."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from mcmc_amd import api
    orc = O.Oracle()
    cores = cpu_cores()
    D = 64
    qd, qo = api.ar1_precision(D, 0.9)
    sweeps = 2048
    t0 = time.perf_counter()
    orc.mvn_chain(qd, qo, [0.0] * D, args.step_size, sweeps, seed=314159265, chain=0)
    per_chain = time.perf_counter() - t0
    chains_per_thread = max(1, int(seconds / per_chain))

    def work(i):
        for c in range(chains_per_thread):
            orc.mvn_chain(qd, qo, [0.0] * D, args.step_size, sweeps, seed=314159265, chain=i * chains_per_thread + c)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": cores * chains_per_thread * D * sweeps / dt, "unit": "proposals/s", "cores": cores,
            "nproc": os.cpu_count(), "kind": "port",
            "sample": f"{cores * chains_per_thread} chains x {D}-D x {sweeps} sweeps (oracle/mcmc_oracle.c orc_mvn_chain, "
                      f"one thread per core, {dt:.1f} s; includes per-sweep trace stores)"}


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
    calls = 40       # ~2 s of the reference's loop on one core (each call returns its 2e6 samples, as the reference does)
    t0 = time.perf_counter()
    for _ in range(calls):
        ref.mh(n, step_size, z, u)
    dt = time.perf_counter() - t0
    return {"value": calls * n / dt, "unit": "proposals/s", "cores": 1, "kind": "reference",
            "sample": f"{calls} calls x {n} iterations of the reference's MetropolisHastingsAlgorithm (FP64, one chain, "
                      f"injected draws: RNG cost excluded), {dt:.2f} s"}


def rng_name():
    """The generator the loaded library was built with (MCMC_HIP_LIB may select the 7-round build)."""
    from mcmc_amd import lib as L
    return f"philox4x32-{L.lib.mcmc_philox_rounds()}"


# ---------------------------------------------------------------------------------------------------
# The three workloads: each returns an object with step() -> float64 statistics tensor (device),
# launch() (the kernel launches alone, for the event bracket), and the constants of its bench line.
class ChainWorkload:
    kernel = "mcmc::chain_kernel"

    def __init__(self, args, rank, world, dev):
        from mcmc_amd import api, lib as L
        self.args, self.api, self.L = args, api, L
        self.rule = {"recip": L.ACCEPT_RECIP, "nodiv": L.ACCEPT_NODIV, "log": L.ACCEPT_LOG, "cross": L.ACCEPT_CROSS}[args.rule]
        self.math = L.MATH_FAST if args.math == "fast" else L.MATH_PRECISE
        self.target = api.make_target(L.TARGET_R2WS, 2.0, 0.54)   # PDF(), metropolis.hpp:19-23 / main.cu:14-19
        self.eng = api.ChainEngine(args.chains, args.coords, self.target, args.step_size, precision=32,
                                   accept_rule=self.rule, math_mode=self.math, n_lags=args.lags, seed=314159265, x0=1.0,
                                   chain_offset=rank * args.chains, center=2.3, device=dev)
        self.proposals_per_launch = args.chains * args.coords * args.sweeps
        self.flop_per_proposal = FLOP_PER_PROPOSAL
        self.algorithmic_bytes = 2 * 4 * args.chains * args.coords        # state load + store

    def launch(self):
        return self.eng.run(self.args.sweeps, window_start=True)

    def config(self, world):
        a = self.args
        return {"workload": f"mh_chain r2WoodsSaxon(2.0,0.54) x{a.coords} coords, {a.chains} chains/GPU, "
                            f"{a.sweeps} sweeps/launch, step {a.step_size}",
                "chains_per_gpu": a.chains, "coords": a.coords, "sweeps_per_step": a.sweeps, "accept_rule": a.rule,
                "math": a.math, "lags_in_kernel": a.lags, "rng": rng_name(),
                "parallelism": f"chains sharded over {world} GPU(s)"}

    # r^2 WS(2.0, 0.54) at width 2.48: mean and variance by quadrature, acceptance from the transition operator of the chain
    # (tests/markov_reference.py, profiles/r3_markov_operator.jsonl) -- independent of any code under test
    KNOWN = {"mean_r": 2.3317236, "var_r": 0.9873116, "acceptance": 0.410037, "tau_int": 5.13564}

    def estimates(self, st, world):
        a = self.args
        s = self.eng.summary(stats_host=st.cpu().numpy(), n_chains=world * a.chains, n_steps=a.sweeps)
        out = {"mean_r": s[0]["mean"], "var_r": s[0]["variance"], "acceptance": s[0]["acceptance"]}
        if a.step_size == 2.48 and a.rule in ("cross", "recip", "nodiv", "log"):
            n = world * a.chains * a.sweeps            # samples of coordinate 0 in the last launch
            se = (self.KNOWN["var_r"] * self.KNOWN["tau_int"] / n) ** 0.5
            out["known_answers"] = dict(self.KNOWN, mean_r_deviation_in_standard_errors=(out["mean_r"] - self.KNOWN["mean_r"]) / se,
                                        source="quadrature and transition operator (tests/markov_reference.py)")
        return out


class NucleonWorkload:
    kernel = "mcmc::xs_chain_kernel (MCIntegrate_R2)"

    def __init__(self, args, rank, world, dev):
        from mcmc_amd import api, lib as L
        self.args, self.api, self.L = args, api, L
        A, c, a = NUCLEI[args.nucleus]
        self.A, self.c, self.a = A, c, a
        self.T = max(1, args.chains // A)                               # events per GPU
        rule = {"recip": L.ACCEPT_RECIP, "nodiv": L.ACCEPT_NODIV, "cross": L.ACCEPT_CROSS}.get(args.rule)
        if rule is None or args.math != "fast":
            raise SystemExit("the nucleon workload runs rules recip / nodiv / cross in fast math")
        self.xs = api.XsEngine(A, 0, self.T, [(L.TARGET_R2WS, c, a)] * A, stepsize=args.step_size, accept_rule=rule,
                               math_mode=L.MATH_FAST, seed=314159265, event_offset=rank * self.T, device=dev)
        self.xs.InitSampleFromPdf()
        self.xs.WarmupMetropolis(64)
        self.proposals_per_launch = self.T * A * args.sweeps
        self.flop_per_proposal = FLOP_PER_PROPOSAL + FLOP_R2_ACCUMULATE
        self.algorithmic_bytes = 3 * 4 * self.T * A + 4 * self.T * A      # state load + store, result store + its row-mean read

    def launch(self):
        _, run_mean = self.xs.MCIntegrate_R2(self.args.sweeps)
        return run_mean          # [A] doubles: mean of r^2 over this rank's events (summed over ranks, / world on the host)

    def config(self, world):
        a = self.args
        return {"workload": f"nucleon radial chains {a.nucleus.upper()}-{self.A} r2WoodsSaxon({self.c},{self.a}), "
                            f"{self.T} events x {self.A} nucleons/GPU, MCIntegrate_R2 {a.sweeps} samples/launch, step {a.step_size}",
                "events_per_gpu": self.T, "nucleons": self.A, "chains_per_gpu": self.T * self.A, "coords": 1,
                "sweeps_per_step": a.sweeps, "accept_rule": a.rule, "math": a.math, "rng": rng_name(),
                "parallelism": f"events sharded over {world} GPU(s)"}

    def estimates(self, st, world):
        r2 = float(st.cpu().numpy().mean()) / world
        return {"rms_fm": r2 ** 0.5, "mean_r2_fm2": r2}


class StressWorkload:
    kernel = "mcmc::mvn_reg_kernel"   # the 64-D configuration runs the register-resident kernel

    def __init__(self, args, rank, world, dev):
        import torch
        from mcmc_amd import api, lib as L
        self.args, self.api, self.L = args, api, L
        self.D, self.lag = 64, 1000
        qd, qo = api.ar1_precision(self.D, 0.9)
        self.eng = api.MvnEngine(args.chains, qd, qo, args.step_size, self.lag, math_mode=L.MATH_FAST, seed=314159265, x0=0.0,
                                 chain_offset=rank * args.chains, device=dev)
        self.obs = torch.empty((args.sweeps, args.chains), dtype=torch.float32, device=dev)
        # --full: the whole state of every sweep, samples_full[sweep][coord][chain] (4 D bytes more per chain-sweep)
        self.full = torch.empty((args.sweeps, self.D, args.chains), dtype=torch.float32, device=dev) if args.full else None
        # fill the lag ring first: a sweep reads its ring slot only from the window's sweep `lag` on, and the timed
        # launches are to carry all 12 B per chain-sweep (observable write, ring read, ring write)
        self.eng.run(self.lag + 24, window=(0, 1 << 40))
        torch.cuda.synchronize()
        self.proposals_per_launch = args.chains * self.D * args.sweeps
        self.flop_per_proposal = FLOP_MVN_PROPOSAL
        # 12 B per chain-sweep (observable write, ring read, ring write) + state load/store per launch
        self.algorithmic_bytes = args.chains * (args.sweeps * (12 + (4 * self.D if args.full else 0)) + 2 * 4 * self.D)
        if args.full:
            self.kernel = "mcmc::mvn_reg_kernel (full-state write-back)"

    def launch(self):
        return self.eng.run(self.args.sweeps, window=(0, 1 << 40), samples_obs=self.obs, samples_full=self.full)

    def config(self, world):
        a = self.args
        wb = "full-state write-back" if a.full else "observable write-back"
        return {"workload": f"stress {self.D}-D Gaussian Sigma=0.9^|i-j|, {a.chains} chains/GPU, lag-{self.lag} ring in HBM, "
                            f"{wb}, {a.sweeps} sweeps/launch, step {a.step_size}",
                "chains_per_gpu": a.chains, "coords": self.D, "sweeps_per_step": a.sweeps, "lag": self.lag,
                "accept_rule": "log", "math": "fast", "rng": rng_name(), "full_state_writeback": bool(a.full),
                "parallelism": f"chains sharded over {world} GPU(s)"}

    def estimates(self, st, world):
        s = st.cpu().numpy()
        n = world * self.args.chains * self.args.sweeps
        return {"mean_x0": s[0] / n, "acceptance": s[2] / (n * self.D)}


WORKLOADS = {"chain": ChainWorkload, "nucleon": NucleonWorkload, "stress": StressWorkload}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args, argv))
    world, rank, local_rank = world_from_env(args)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    backend = os.environ.get("MCMC_BENCH_BACKEND", "nccl")   # "nccl" is RCCL; "gloo" for a one-GPU rehearsal
    n_dev = max(torch.cuda.device_count(), 1)
    if world > n_dev and backend == "nccl":
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, this node shows {n_dev} "
                         f"(MCMC_BENCH_BACKEND=gloo rehearses more ranks than GPUs)")
    local_rank %= n_dev      # only matters when more ranks than GPUs are rehearsed on one card
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group of {dist.get_world_size()} ranks, --gpus {args.gpus}")
    n_gpus = dist.get_world_size() if dist is not None else 1

    trace_on = os.environ.get("MCMC_BENCH_TRACE") == "1"

    def trace(msg):
        if trace_on:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    trace("process group up")
    from mcmc_amd import api, dist as D, lib as L
    wl = WORKLOADS[args.workload](args, rank, world, dev)

    def step():
        st = wl.launch()
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
        st = wl.launch()
        evs[i][1].record()
        D.allreduce_stats(st)
    fence()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, device=dev)
    trace(f"timed region done: {elapsed:.3f} s")
    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / args.steps

    value = n_gpus * wl.proposals_per_launch * args.steps / elapsed
    estimates = wl.estimates(st, n_gpus)

    out = None
    if rank == 0:
        launch_props = wl.proposals_per_launch
        achieved = wl.flop_per_proposal * launch_props / (kernel_ms * 1e-3) / 1e12
        roof = {"bound": "valu", "achieved": achieved, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / VALU_PEAK_TFLOPS, "traffic": None, "kernel": wl.kernel, "kernel_ms": kernel_ms,
                "flop_per_proposal": wl.flop_per_proposal, "proposals_per_launch": launch_props,
                "algorithmic_bytes_per_launch": wl.algorithmic_bytes,
                "hbm": {"achieved_GBps": wl.algorithmic_bytes / (kernel_ms * 1e-3) / 1e9, "peak_GBps": HBM_PEAK_GBPS,
                        "frac": wl.algorithmic_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}}
        if args.workload == "chain":
            # HBM bytes per launch: NOT measured in this run -- the committed PMC passes of this same workload
            # (tools/gpu_profile.sh; FETCH_SIZE / WRITE_SIZE in their own runs, FETCH doubled per MI355X_MICROARCH.md)
            for prof in ("r3_pmc_chain_kernel.json", "r2_pmc_chain_kernel.json"):   # this round's passes, else the round before's
                try:
                    pm = json.load(open(os.path.join(ROOT, "profiles", prof)))
                    if args.chains == 1 << 24 and args.coords == 3 and args.lags == 0:
                        roof["traffic"] = pm["derived"]["hbm_bytes_per_launch"]
                        roof["traffic_source"] = (f"profiles/{prof} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                                  "this workload, tools/gpu_profile.sh; not measured in this run)")
                    break
                except (OSError, KeyError, ValueError):
                    pass
            roof["note"] = ("FP32 VALU-issue-bound (state in registers, no per-step HBM traffic). achieved = 24 algorithmic "
                            "FP32 flop per proposal (SURVEY 8d) / kernel time; Philox integer work and quarter-rate "
                            "transcendentals are not counted as flops, so frac understates VALU occupancy: valu_issue sets "
                            "the ideal issue cycles of the loop's instruction mix (2 / 4 / 8 per class) against the cycles "
                            "measured with the shader clock of THIS run (DESIGN.md 4.1)")
        if args.workload in ("nucleon", "stress"):
            # HBM bytes per launch of this workload's kernel from its committed PMC passes (not measured in this run)
            prof = {"nucleon": "r2_pmc_nucleon_kernel.json", "stress": "r2_pmc_stress_kernel.json"}[args.workload]
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", prof)))
                if pm["derived"].get("algorithmic_bytes_per_launch") == wl.algorithmic_bytes:   # same shape as profiled
                    roof["traffic"] = pm["derived"]["hbm_bytes_per_launch"]
                    roof["traffic_source"] = (f"profiles/{prof} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                              "workload, tools/gpu_profile.sh; not measured in this run)")
            except (OSError, KeyError, ValueError):
                pass
        if args.workload == "stress" and args.full:
            # the HBM-heavy form of configs[4]: 268 B per chain-sweep; the line's roofline is the HBM one
            # (the VALU figures stay beside it: the kernel overlaps its stores with 64 proposals per chain-sweep)
            gbps = wl.algorithmic_bytes / (kernel_ms * 1e-3) / 1e9
            roof.update({"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                         "valu": {"achieved_TFLOPs": achieved, "peak_TFLOPs": VALU_PEAK_TFLOPS, "frac": achieved / VALU_PEAK_TFLOPS}})
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r3_pmc_stress_full_kernel.json")))
                if pm["derived"].get("algorithmic_bytes_per_launch") == wl.algorithmic_bytes:
                    roof["traffic"] = pm["derived"]["hbm_bytes_per_launch"]
                    roof["traffic_source"] = ("profiles/r3_pmc_stress_full_kernel.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                              "of this workload, tools/gpu_profile.sh; not measured in this run)")
            except (OSError, KeyError, ValueError):
                pass
        if args.workload == "nucleon":
            roof["note"] = ("FP32 VALU-issue-bound like the chain kernel; 24 + 6 (Kahan r^2) algorithmic flop per proposal. "
                            "kernel_ms brackets xs_chain_kernel + the two row-mean kernels of the launch")
        elif args.workload == "stress" and args.full:
            roof["note"] = ("configs[4] with every sample written back: 12 + 4 x 64 B of HBM traffic per chain-sweep (observable, lag "
                            "ring read + write, the 64-D state) + state load/store per launch, against 8 TB/s; `valu` gives the "
                            "22-flop-per-update figure of the same launch")
        elif args.workload == "stress":
            roof["note"] = ("BASELINE configs[4] calls this the HBM-bound path; measured it is VALU-issue-bound: 64 proposals "
                            "per 12 B of HBM traffic per chain-sweep. `hbm` gives the algorithmic bytes (12 B per chain-sweep + "
                            "state load/store) against 8 TB/s; bound and frac are the VALU figures (22 flop per single-"
                            "coordinate update)")
        out = {
            "metric": "MH proposals/sec", "value": value, "unit": "proposals/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": wl.config(n_gpus), "roofline": roof, "estimates": estimates,
            "collective": (f"1 all_reduce(SUM, float64[{st.numel()}]) per step, backend "
                           f"{'rccl' if backend == 'nccl' else backend}" if n_gpus > 1 else None),
        }

    # Shader clock this kernel sustains, measured in THIS run: the same kernel body with s_memtime / s_memrealtime
    # stamps around each block's tile loop (diagnostic instantiation, mcmc_chain_clock_probe), after the timed region.
    if args.workload == "chain" and not args.no_clock and args.math == "fast" and args.rule in ("cross", "recip"):
        try:
            for _ in range(3):
                wl.launch()
            ghz = api.chain_clock_probe(wl.eng, args.sweeps)
        except Exception as e:
            ghz = None
            trace(f"clock probe failed: {e}")
        if out is not None and ghz is not None:
            valu = {"clock_GHz": ghz, "clock_source": "s_memtime / s_memrealtime stamps in this run (mcmc_chain_clock_probe)"}
            cyc = 1024 * ghz * 1e9 * 64 * (kernel_ms * 1e-3) / wl.proposals_per_launch
            valu["measured_cycles_per_wave_proposal"] = cyc
            mix_file = "r2_chain_kernel_isa_mix.json" if os.path.exists(os.path.join(ROOT, "profiles", "r2_chain_kernel_isa_mix.json")) \
                else "r1_chain_kernel_isa_mix.json"
            try:
                mix = json.load(open(os.path.join(ROOT, "profiles", mix_file)))
                if args.rule == "cross" and args.lags == 0:
                    valu.update({"ideal_issue_cycles": mix["issue_cycles_per_wave_proposal_ideal"],
                                 "frac_of_ideal_issue": mix["issue_cycles_per_wave_proposal_ideal"] / cyc,
                                 "valu_insts_per_proposal": mix["valu_insts_per_proposal"],
                                 "isa_mix_source": f"profiles/{mix_file} (static, tools/isa_mix.py; 2 / 4 / 8 cycles per class)"})
            except (OSError, KeyError, ValueError):
                pass
            out["roofline"]["valu_issue"] = valu

    # ESS/sec: tau_int = 1 + 2 sum rho_k from the streaming lag-49 accumulators at the same width, on a chain
    # subset after warm-up (outside the timed region); the same launches give the throughput of the kernel
    # with all 49 lag products carried by every chain.
    if args.workload == "chain" and not args.no_ess:
        sub = min(args.chains, 1 << 22)
        e2 = api.ChainEngine(sub, args.coords, wl.target, args.step_size, precision=32, accept_rule=wl.rule, math_mode=wl.math,
                             n_lags=L.MAX_LAGS, seed=314159265, x0=1.0, chain_offset=rank * args.chains, center=2.3,
                             device=dev)
        e2.run(512, n_lags=0)
        e2.run(256, window_start=True, n_lags=L.MAX_LAGS)
        torch.cuda.synchronize()
        rates = {}
        lag_launches = {1024: 4, 4096: 2}
        for lag_sweeps, reps in lag_launches.items():   # a step-size study's launches are 10^3 - 10^4 sweeps (C2: 10^4),
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # one after the other
            e2.run(lag_sweeps, window_start=True, n_lags=L.MAX_LAGS)   # untimed warm-up launch, no idle gap before the timed ones
            ea.record()
            for _ in range(reps):
                e2.run(lag_sweeps, window_start=True, n_lags=L.MAX_LAGS)
            eb.record()
            torch.cuda.synchronize()
            rates[lag_sweeps] = reps * sub * args.coords * lag_sweeps / (ea.elapsed_time(eb) * 1e-3)
        tau = e2.summary()[0]["tau_int"]
        if out is not None:
            out["tau_int"] = tau
            # two readings of "ESS/sec": the headline kernel's rate (mean / variance / acceptance only) over tau_int, and the
            # rate of the kernel that MEASURES tau_int (all 49 lag products of every chain, on the matrix cores) over it
            out["ess_per_sec"] = value / tau
            out["lag49_proposals_per_sec_per_gpu"] = rates[4096]
            out["lag49_sweeps_per_launch"] = 4096
            out["lag49_launches_timed"] = lag_launches   # back to back between two events; one statistics window per launch
            out["lag49_proposals_per_sec_per_gpu_1024_sweeps"] = rates[1024]
            out["ess_per_sec_lag49_kernel"] = n_gpus * rates[4096] / tau
            out["lag49_kernel"] = "mcmc::chain_lag_kernel (v_mfma_f32_16x16x32_f16)" if os.environ.get("MCMC_LAG_MFMA", "1") != "0" \
                else "mcmc::chain_kernel<NLAG=49> (VALU)"
    if rank == 0 and n_gpus == 1 and args.workload == "chain" and not args.no_secondary:
        out["secondary"] = secondary(dev)
        p7 = philox7_rate(args)
        if p7 is not None:
            out["secondary"]["chain_philox4x32_7_proposals_per_s"] = p7
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        if args.workload == "chain":
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
            out["cpu_reference_f64"] = cpu_reference_f64(args.step_size)
        elif args.workload == "nucleon":
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds, target=(wl.c, wl.a),
                                               note="; the chain step only, without the Kahan r^2 accumulation")
        else:
            out["cpu_baseline"] = cpu_baseline_mvn(args, args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
