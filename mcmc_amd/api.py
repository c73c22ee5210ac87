"""Host-side mirror of the reference's driver surface on top of the C ABI.

Names follow the reference: `MetropolisHastingsAlgorithm` (stepsize/include/
metropolis.hpp:28), `WarmupMetropolis` / `MCIntegrate_R2` / `MCIntegrate_S_AB`
(xsection/include/*.cuh), `Average` / `Variance` / `AutoCorrelation`
(Statistics.cuh, statistics.hpp).  torch tensors are used only as device
buffers; every computation happens inside libmcmc_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L
from .lib import lib as _lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def make_target(kind, p0, p1=0.0):
    return L.Target(int(kind), float(p0), float(p1))


class ChainEngine:
    """n_chains x n_coords independent 1-D MH chains with streaming statistics
    (Layer A.1, `mcmc_chain_run`).  Owns state / stats / workspace tensors so a
    step allocates nothing."""

    def __init__(self, n_chains, n_coords, target, step, precision=32, accept_rule=L.ACCEPT_RECIP,
                 math_mode=L.MATH_FAST, n_lags=0, seed=314159265, x0=1.0, chain_offset=0, center=0.0,
                 device="cuda:0", carry_ring=False):
        self.device = torch.device(device)
        self.n_chains, self.n_coords, self.n_lags = int(n_chains), int(n_coords), int(n_lags)
        self.dtype = torch.float64 if precision == 64 else torch.float32
        self.state = torch.full((n_coords, n_chains), float(x0), dtype=self.dtype, device=self.device)
        self.stats = torch.zeros((n_coords, L.stats_len(n_lags)), dtype=torch.float64, device=self.device)
        self.ring = (torch.zeros((n_coords, max(n_lags, 1), n_chains), dtype=self.dtype, device=self.device)
                     if carry_ring and n_lags > 0 else None)
        self.args = L.ChainArgs()
        a = self.args
        a.n_chains, a.n_coords, a.n_lags = n_chains, n_coords, n_lags
        a.step_size, a.center = float(step), float(center)
        a.target = target
        a.precision, a.accept_rule, a.math_mode = precision, accept_rule, math_mode
        a.seed, a.step_offset, a.chain_offset = seed, 0, chain_offset
        a.state, a.stats = self.state.data_ptr(), self.stats.data_ptr()
        a.ring = self.ring.data_ptr() if self.ring is not None else None
        nbytes = _lib.mcmc_chain_workspace_bytes(C.byref(a))
        self.workspace = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=self.device)
        a.workspace, a.workspace_bytes = self.workspace.data_ptr(), nbytes
        self.steps_done = 0
        self.window_steps = 0

    def run(self, n_steps, window_start=True, samples=None, inj_normals=None, inj_uniforms=None, accept_bits=None,
            n_lags=None, noalloc=False):
        """Launch n_steps sweeps (asynchronous on the current torch stream).
        Returns the device stats tensor [n_coords][stats_len].  noalloc: through `mcmc_chain_run_noalloc`
        (the library's static scratch instead of this engine's workspace)."""
        a = self.args
        a.n_steps = int(n_steps)
        a.window_start = 1 if window_start else 0
        a.step_offset = self.steps_done
        a.samples = _p(samples)
        a.inj_normals = _p(inj_normals)
        a.inj_uniforms = _p(inj_uniforms)
        a.accept_bits = _p(accept_bits)
        if n_lags is not None:
            a.n_lags = int(n_lags)
        if noalloc:
            ws, a.workspace, a.workspace_bytes = (a.workspace, a.workspace_bytes), None, 0   # prove it is not used
            try:
                L.check(_lib.mcmc_chain_run_noalloc(C.byref(a), _stream()))
            finally:
                a.workspace, a.workspace_bytes = ws
        else:
            L.check(_lib.mcmc_chain_run(C.byref(a), _stream()))
        self.steps_done += int(n_steps)
        self.window_steps = int(n_steps) if window_start else self.window_steps + int(n_steps)
        return self.stats

    def summary(self, stats_host=None, n_chains=None, n_steps=None):
        """Finalise (host) the statistics of each coordinate -> list of dicts."""
        st = self.stats.cpu().numpy() if stats_host is None else np.asarray(stats_host, dtype=np.float64)
        return [finalize_stats(st[c], self.args.n_lags, n_chains or self.n_chains, n_steps or self.window_steps,
                               self.args.center) for c in range(st.shape[0])]


def chain_clock_probe(engine, n_steps, n_lags=None):
    """`mcmc_chain_clock_probe`: shader clock (GHz) the chain kernel of `engine` sustains over one launch of
    n_steps sweeps (stamped twin of the kernel; advances the chains)."""
    a = engine.args
    a.n_steps, a.window_start, a.step_offset = int(n_steps), 1, engine.steps_done
    a.samples = a.inj_normals = a.inj_uniforms = a.accept_bits = None
    if n_lags is not None:
        a.n_lags = int(n_lags)
    ghz = C.c_double()
    L.check(_lib.mcmc_chain_clock_probe(C.byref(a), C.byref(ghz), _stream()))
    engine.steps_done += int(n_steps)
    engine.window_steps = int(n_steps)
    return ghz.value


def finalize_stats(stats_row, n_lags, n_chains, n_steps, center):
    row = np.ascontiguousarray(stats_row, dtype=np.float64)
    out = L.ChainSummary()
    rho = (C.c_double * max(n_lags, 1))()
    L.check(_lib.mcmc_stats_finalize(row.ctypes.data_as(C.POINTER(C.c_double)), n_lags, n_chains, n_steps,
                                     float(center), C.byref(out), rho))
    return {"mean": out.mean, "variance": out.variance, "acceptance": out.acceptance, "tau_int": out.tau_int,
            "rho": np.array(rho[:n_lags], dtype=np.float64)}


def MetropolisHastingsAlgorithm(iterations, stepSize, target=None, n_chains=1, precision=64,
                                accept_rule=L.ACCEPT_SHORTCIRCUIT, math_mode=L.MATH_PRECISE, seed=314159265,
                                x0=1.0, n_lags=0, n_warmup=0, return_samples=True):
    """stepsize/include/metropolis.hpp:28 for n_chains chains on the GPU.
    Returns (acceptance ratio, samples[iterations][n_chains]) like the reference's
    pair<double, vector<double>>, plus the finalised summary dict."""
    target = target or make_target(L.TARGET_R2WS, 2.0, 0.54)  # PDF(), metropolis.hpp:19-23
    summ = (L.ChainSummary * 1)()
    rho = (C.c_double * max(n_lags, 1))()
    dt = np.float64 if precision == 64 else np.float32
    samples = np.empty((iterations, 1, n_chains), dtype=dt) if return_samples else None
    L.check(_lib.mcmc_stepsize_run(C.byref(target), precision, accept_rule, math_mode, float(stepSize), n_chains, 1,
                                   n_warmup, iterations, n_lags, seed, float(x0), summ, rho,
                                   samples.ctypes.data_as(C.c_void_p) if samples is not None else None))
    s = summ[0]
    info = {"mean": s.mean, "variance": s.variance, "acceptance": s.acceptance, "tau_int": s.tau_int,
            "rho": np.array(rho[:n_lags])}
    return s.acceptance, (samples[:, 0, :] if samples is not None else None), info


def TuneStepSize(target, lo, hi, n_widths=9, n_chains=1 << 16, n_warmup=1000, n_steps=2000, seed=314159265, x0=1.0):
    """`mcmc_tune_stepsize`: the stepsize executable's study on the device -> (best_width, widths, acceptance, tau_int)."""
    w, a, t = (C.c_double * n_widths)(), (C.c_double * n_widths)(), (C.c_double * n_widths)()
    best = C.c_double()
    L.check(_lib.mcmc_tune_stepsize(C.byref(target), float(lo), float(hi), int(n_widths), int(n_chains), int(n_warmup),
                                    int(n_steps), int(seed), float(x0), w, a, t, C.byref(best)))
    return best.value, np.array(w[:]), np.array(a[:]), np.array(t[:])


class XsEngine:
    """xsection Layer A on torch buffers: prevSample[nN][T], per-nucleon targets,
    result buffers and workspace; methods mirror the reference kernels."""

    def __init__(self, nA, nB, T, targets, stepsize=2.2, accept_rule=L.ACCEPT_RECIP, math_mode=L.MATH_FAST,
                 seed=314159265, event_offset=0, sigmaNN=4.3, beta=0.2, init=1.0, device="cuda:0"):
        self.device = torch.device(device)
        self.nA, self.nB, self.nN, self.T = nA, nB, nA + nB, int(T)
        assert len(targets) == self.nN
        tg = np.zeros(self.nN, dtype=np.dtype([("kind", "<i4"), ("p0", "<f4"), ("p1", "<f4")]))
        for i, (k, p0, p1) in enumerate(targets):
            tg[i] = (k, p0, p1)
        kinds = set(int(k) for k, _, _ in targets)
        self.targets = torch.from_numpy(tg.view(np.uint8).copy()).to(self.device)
        self.prevSample = torch.empty((self.nN, self.T), dtype=torch.float32, device=self.device)
        self.resultR2 = torch.empty((self.nN, self.T), dtype=torch.float32, device=self.device)
        self.resultSAB = torch.empty((self.T,), dtype=torch.float32, device=self.device)
        self.runMean = torch.zeros((self.nN,), dtype=torch.float64, device=self.device)
        self.sabMean = torch.zeros((1,), dtype=torch.float64, device=self.device)
        self.hits = torch.zeros((4,), dtype=torch.float64, device=self.device)
        a = self.args = L.XsArgs()
        a.nNucleonsA, a.nNucleonsB, a.totalThreads = nA, nB, self.T
        a.targets, a.prevSample = self.targets.data_ptr(), self.prevSample.data_ptr()
        a.stepsize, a.accept_rule, a.math_mode = float(stepsize), accept_rule, math_mode
        a.uniform_kind = kinds.pop() if len(kinds) == 1 else -1
        a.seed, a.step_offset, a.event_offset = seed, 0, event_offset
        a.sigmaNN, a.beta = float(sigmaNN), float(beta)
        nbytes = _lib.mcmc_xs_workspace_bytes(C.byref(a))
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        a.workspace, a.workspace_bytes = self.workspace.data_ptr(), nbytes
        self.InitSampleArray(init)

    def _inject(self, injected):
        a = self.args
        if injected is None:
            a.injected, a.injected_stride = None, 0
        else:
            assert injected.dtype == torch.float32 and injected.is_contiguous() and injected.shape[0] == self.T
            a.injected, a.injected_stride = injected.data_ptr(), injected.shape[1]

    def InitSampleArray(self, initSample):
        L.check(_lib.mcmc_launch_init_sample_array(_p(self.prevSample), float(initSample), self.nN, self.T, _stream()))

    def InitSampleFromPdf(self):
        """Randomised starts drawn from each nucleon's tabulated density (main.cu:144-148's suggestion)."""
        L.check(_lib.mcmc_launch_init_sample_from_pdf(C.byref(self.args), _stream()))

    def WarmupMetropolis(self, iterations, stepsize=None, injected=None):
        if stepsize is not None:
            self.args.stepsize = float(stepsize)
        self._inject(injected)
        L.check(_lib.mcmc_launch_warmup_metropolis(C.byref(self.args), int(iterations), _stream()))
        self.args.step_offset += int(iterations)

    def MCIntegrate_R2(self, samples, stepsize=None, injected=None):
        if stepsize is not None:
            self.args.stepsize = float(stepsize)
        self._inject(injected)
        L.check(_lib.mcmc_launch_integrate_r2(C.byref(self.args), _p(self.resultR2), int(samples), _p(self.runMean),
                                              _stream()))
        self.args.step_offset += int(samples)
        return self.resultR2, self.runMean

    def MCIntegrate_S_AB(self, samples, impactParameter, stepsize=None, injected=None, dmax2=None):
        if stepsize is not None:
            self.args.stepsize = float(stepsize)
        self._inject(injected)
        L.check(_lib.mcmc_launch_integrate_sab(C.byref(self.args), _p(self.resultSAB), int(samples),
                                               float(impactParameter), _p(self.sabMean),
                                               _p(self.hits) if dmax2 is not None else None,
                                               float(dmax2 if dmax2 is not None else 0.0), _stream()))
        self.args.step_offset += int(samples)
        return self.resultSAB, self.sabMean, (self.hits if dmax2 is not None else None)

    def MCIntegrate_Inel(self, samples, impactParameter, dmax2, stepsize=None, injected=None):
        """Black-disk inelastic flag only (`mcmc_launch_integrate_inel`): returns the device vector
        [samples with >= 1 collision, samples]."""
        if stepsize is not None:
            self.args.stepsize = float(stepsize)
        self._inject(injected)
        L.check(_lib.mcmc_launch_integrate_inel(C.byref(self.args), int(samples), float(impactParameter),
                                                _p(self.hits), float(dmax2), _stream()))
        self.args.step_offset += int(samples)
        return self.hits[:2]


def ar1_precision(dim, rho):
    """Tridiagonal precision matrix of Sigma_ij = rho^|i-j| -> (q_diag[dim], q_off[dim-1])."""
    c = 1.0 / (1.0 - rho * rho)
    d = np.full(dim, (1.0 + rho * rho) * c, dtype=np.float32)
    d[0] = d[-1] = c
    return d, np.full(dim - 1, -rho * c, dtype=np.float32)


class MvnEngine:
    """Layer A.3 (`mcmc_mvn_run`): chains on a correlated Gaussian with tridiagonal precision, lag-L
    autocorrelation of one coordinate through an HBM ring, optional sample write-back."""

    def __init__(self, n_chains, q_diag, q_off, step, lag, observable=0, math_mode=L.MATH_FAST, seed=314159265,
                 x0=1.0, chain_offset=0, center=0.0, device="cuda:0"):
        self.device = torch.device(device)
        self.q_diag = np.ascontiguousarray(q_diag, dtype=np.float32)
        self.q_off = np.ascontiguousarray(q_off, dtype=np.float32)
        self.dim, self.n_chains, self.lag = len(self.q_diag), int(n_chains), int(lag)
        self.state = torch.full((self.dim, self.n_chains), float(x0), dtype=torch.float32, device=self.device)
        self.ring = torch.zeros((self.lag, self.n_chains), dtype=torch.float32, device=self.device)
        self.stats = torch.zeros((6,), dtype=torch.float64, device=self.device)
        a = self.args = L.MvnArgs()
        a.n_chains, a.dim, a.observable = self.n_chains, self.dim, observable
        a.step_size, a.center = float(step), float(center)
        a.q_diag, a.q_off = self.q_diag.ctypes.data, self.q_off.ctypes.data
        a.math_mode, a.lag = math_mode, self.lag
        a.seed, a.step_offset, a.chain_offset = seed, 0, chain_offset
        a.state, a.ring, a.stats = self.state.data_ptr(), self.ring.data_ptr(), self.stats.data_ptr()
        nbytes = _lib.mcmc_mvn_workspace_bytes(C.byref(a))
        self.workspace = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=self.device)
        a.workspace, a.workspace_bytes = self.workspace.data_ptr(), nbytes
        self.steps_done = 0

    def run(self, n_steps, window=None, samples_obs=None, samples_full=None, inj=None, accept_bits=None):
        """n_steps sweeps; window = (begin, end) global sweep indices of the statistics window (default: this
        launch alone).  Returns the device stats tensor [6] of THIS launch."""
        a = self.args
        a.n_steps, a.step_offset = int(n_steps), self.steps_done
        wb, we = window if window is not None else (self.steps_done, self.steps_done + int(n_steps))
        a.window_begin, a.window_end = int(wb), int(we)
        a.samples_obs, a.samples_full, a.inj, a.accept_bits = _p(samples_obs), _p(samples_full), _p(inj), _p(accept_bits)
        L.check(_lib.mcmc_mvn_run(C.byref(a), _stream()))
        self.steps_done += int(n_steps)
        return self.stats

    def summary(self, stats_host, n_sweeps, n_chains=None):
        st = np.ascontiguousarray(stats_host, dtype=np.float64)
        out, rho = L.ChainSummary(), C.c_double()
        L.check(_lib.mcmc_mvn_finalize(st.ctypes.data_as(C.POINTER(C.c_double)), n_chains or self.n_chains, self.dim,
                                       int(n_sweeps), self.lag, float(self.args.center), C.byref(out), C.byref(rho)))
        return {"mean": out.mean, "variance": out.variance, "acceptance": out.acceptance, "rho_lag": rho.value}


# --- host statistics with the reference's arithmetic -------------------------
def _f32(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return x, x.ctypes.data_as(C.POINTER(C.c_float))


def _f64(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return x, x.ctypes.data_as(C.POINTER(C.c_double))


def Sum(samples):
    x, p = _f32(samples)
    return _lib.mcmc_Sum(p, len(x))


def Average(samples):
    """Statistics.cuh:23 for float32 input, statistics.hpp:6 for float64 input."""
    if np.asarray(samples).dtype == np.float64:
        x, p = _f64(samples)
        return _lib.mcmc_Average_f64(p, len(x))
    x, p = _f32(samples)
    return _lib.mcmc_Average(p, len(x))


def Variance(samples, avg=None):
    """Statistics.cuh:28 (population, FP32 Kahan; :45 when avg is None) or statistics.hpp:30 (n-1, FP64)."""
    if np.asarray(samples).dtype == np.float64:
        x, p = _f64(samples)
        return _lib.mcmc_Variance_f64(p, len(x))
    x, p = _f32(samples)
    if avg is None:
        return _lib.mcmc_Variance_n(p, len(x))
    return _lib.mcmc_Variance(p, float(avg), len(x))


def AutoCorrelation(samples, lag, avg=None):
    """Statistics.cuh:63 (FP32) or statistics.hpp:53 (FP64)."""
    if np.asarray(samples).dtype == np.float64:
        x, p = _f64(samples)
        return _lib.mcmc_AutoCorrelation_f64(p, len(x), int(lag))
    x, p = _f32(samples)
    if avg is None:
        return _lib.mcmc_AutoCorrelation_n(p, int(lag), len(x))   # Statistics.cuh:90
    return _lib.mcmc_AutoCorrelation(p, float(avg), int(lag), len(x))


def CrossSections(b, S_AB, S_AB_stderr):
    """xsection/tools/XSectionAnalysis.py:10-28 -> dict of (value, uncertainty) in mb."""
    b, pb = _f64(b)
    S, pS = _f64(S_AB)
    E, pE = _f64(S_AB_stderr)
    out = (C.c_double * 6)()
    L.check(_lib.mcmc_cross_sections(pb, pS, pE, len(b), out))
    return {"tot": (out[0], out[1]), "el": (out[2], out[3]), "rxn": (out[4], out[5])}
