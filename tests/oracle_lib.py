"""ctypes binding of the CPU oracle (oracle/liboracle.so) and, when built, of the
reference's own stepsize headers (oracle/_ref/libref_stepsize.so).

TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (mcmc_amd) never imports it.
"""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE = os.path.join(_ROOT, "oracle", "liboracle.so")
_REF = os.path.join(_ROOT, "oracle", "_ref", "libref_stepsize.so")

WS, R2WS, GAUSS, R2GAUSS = 0, 1, 2, 3


class OrcTarget(C.Structure):
    _fields_ = [("kind", C.c_int), ("p0", C.c_float), ("p1", C.c_float)]


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


def load_oracle():
    if not os.path.exists(_ORACLE):
        raise RuntimeError(f"{_ORACLE} missing: run `make -C oracle` (or __graft_entry__.build())")
    lib = C.CDLL(_ORACLE)
    lib.orc_average_f64.restype = C.c_double
    lib.orc_variance_f64.restype = C.c_double
    lib.orc_autocorrelation_f64.restype = C.c_double
    lib.orc_sum_f32.restype = C.c_float
    lib.orc_average_f32.restype = C.c_float
    lib.orc_variance_f32.restype = C.c_float
    lib.orc_autocorrelation_f32.restype = C.c_float
    lib.orc_variance_n_f32.restype = C.c_float
    lib.orc_autocorrelation_n_f32.restype = C.c_float
    lib.orc_target_f32.restype = C.c_float
    lib.orc_target_f64.restype = C.c_double
    lib.orc_R2WoodsSaxon_f64.restype = C.c_double
    lib.orc_R2Gaussian_f64.restype = C.c_double
    lib.orc_GammaNNR2.restype = C.c_float
    lib.orc_GammaNN.restype = C.c_float
    lib.orc_integrate_sab_thread.restype = C.c_float
    lib.orc_baseline_chains_f32.restype = C.c_uint64
    lib.orc_mvn_chain.restype = C.c_uint64
    return lib


def have_ref():
    return os.path.exists(_REF)


def load_ref():
    lib = C.CDLL(_REF)
    lib.ref_MetropolisHastingsAlgorithm.restype = C.c_double
    for f in ("ref_R2WoodsSaxon", "ref_R2Gaussian", "ref_PDF", "ref_Average", "ref_Variance", "ref_AutoCorrelation"):
        getattr(lib, f).restype = C.c_double
    return lib


class Oracle:
    def __init__(self):
        self.lib = load_oracle()

    # --- RNG -----------------------------------------------------------
    def set_philox_rounds(self, rounds):
        self.lib.orc_set_philox_rounds(C.c_int(int(rounds)))

    def philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        self.lib.orc_philox4x32_10(c, k, o)
        return list(o)

    def mh_draws_f32(self, seed, chain, coord, t):
        n, u = C.c_float(), C.c_float()
        self.lib.orc_mh_draws_f32(C.c_uint64(seed), C.c_uint64(chain), C.c_uint32(coord), C.c_uint64(t),
                                  C.byref(n), C.byref(u))
        return n.value, u.value

    def sab_mh_draws_f32(self, seed, chain, coord, t):
        n, u = C.c_float(), C.c_float()
        self.lib.orc_sab_mh_draws_f32(C.c_uint64(seed), C.c_uint64(chain), C.c_uint32(coord), C.c_uint64(t),
                                      C.byref(n), C.byref(u))
        return n.value, u.value

    def angle_draws_f32(self, seed, chain, coord, t):
        a, b = C.c_float(), C.c_float()
        self.lib.orc_angle_draws_f32(C.c_uint64(seed), C.c_uint64(chain), C.c_uint32(coord), C.c_uint64(t),
                                     C.byref(a), C.byref(b))
        return a.value, b.value

    def mh_draws_f64(self, seed, chain, coord, t):
        n, u = C.c_double(), C.c_double()
        self.lib.orc_mh_draws_f64(C.c_uint64(seed), C.c_uint64(chain), C.c_uint32(coord), C.c_uint64(t),
                                  C.byref(n), C.byref(u))
        return n.value, u.value

    # --- a1 -------------------------------------------------------------
    def mh_chain_f64(self, iterations, step, kind, p0, p1, x0, normals, uniforms):
        normals = np.ascontiguousarray(normals, dtype=np.float64)
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
        samples = np.empty(iterations, dtype=np.float64)
        bits = np.empty(iterations, dtype=np.uint8)
        used = C.c_int()
        acc = self.lib.orc_mh_chain_f64(C.c_int(iterations), C.c_double(step), C.c_int(kind), C.c_double(p0),
                                        C.c_double(p1), C.c_double(x0), _ptr(normals, C.c_double),
                                        _ptr(uniforms, C.c_double), _ptr(samples, C.c_double),
                                        _ptr(bits, C.c_uint8), C.byref(used))
        return acc, samples, bits, used.value

    def mh_chain_f64_philox(self, iterations, step, kind, p0, p1, x0, seed, chain, coord, offset):
        samples = np.empty(iterations, dtype=np.float64)
        bits = np.empty(iterations, dtype=np.uint8)
        acc = self.lib.orc_mh_chain_f64_philox(C.c_int(iterations), C.c_double(step), C.c_int(kind), C.c_double(p0),
                                               C.c_double(p1), C.c_double(x0), C.c_uint64(seed), C.c_uint64(chain),
                                               C.c_uint32(coord), C.c_uint64(offset), _ptr(samples, C.c_double),
                                               _ptr(bits, C.c_uint8))
        return acc, samples, bits

    # --- a4 / a14 ---------------------------------------------------------
    def average_f64(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return self.lib.orc_average_f64(_ptr(x, C.c_double), C.c_int(len(x)))

    def variance_f64(self, x, avg=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if avg is None:
            avg = self.average_f64(x)
        return self.lib.orc_variance_f64(_ptr(x, C.c_double), C.c_int(len(x)), C.c_double(avg))

    def autocorrelation_f64(self, x, lag):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return self.lib.orc_autocorrelation_f64(_ptr(x, C.c_double), C.c_int(len(x)), C.c_int(lag))

    def sum_f32(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        return self.lib.orc_sum_f32(_ptr(x, C.c_float), C.c_int(len(x)))

    def average_f32(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        return self.lib.orc_average_f32(_ptr(x, C.c_float), C.c_int(len(x)))

    def variance_f32(self, x, avg):
        x = np.ascontiguousarray(x, dtype=np.float32)
        return self.lib.orc_variance_f32(_ptr(x, C.c_float), C.c_float(avg), C.c_int(len(x)))

    def autocorrelation_f32(self, x, avg, lag):
        x = np.ascontiguousarray(x, dtype=np.float32)
        return self.lib.orc_autocorrelation_f32(_ptr(x, C.c_float), C.c_float(avg), C.c_int(lag), C.c_int(len(x)))

    def variance_n_f32(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        return self.lib.orc_variance_n_f32(_ptr(x, C.c_float), C.c_int(len(x)))

    def autocorrelation_n_f32(self, x, lag):
        x = np.ascontiguousarray(x, dtype=np.float32)
        return self.lib.orc_autocorrelation_n_f32(_ptr(x, C.c_float), C.c_int(lag), C.c_int(len(x)))

    # --- a5-a12 -----------------------------------------------------------
    @staticmethod
    def _targets(targets):
        arr = (OrcTarget * len(targets))()
        for i, (k, p0, p1) in enumerate(targets):
            arr[i].kind, arr[i].p0, arr[i].p1 = int(k), float(p0), float(p1)
        return arr

    def warmup_thread(self, targets, x, step, iterations, draws):
        nN = len(targets)
        x = np.array(x, dtype=np.float32)
        draws = np.ascontiguousarray(draws, dtype=np.float32)
        assert draws.size >= iterations * nN * 2
        bits = np.empty(iterations * nN, dtype=np.uint8)
        self.lib.orc_warmup_thread(C.c_int(nN), self._targets(targets), _ptr(x, C.c_float), C.c_float(step),
                                   C.c_int(iterations), _ptr(draws, C.c_float), _ptr(bits, C.c_uint8))
        return x, bits.reshape(iterations, nN)

    def warmup_thread_trace(self, targets, x, step, iterations, draws):
        """-> final x, accept bits [it][nN], x after every sweep [it][nN], min |p'*invP - U|."""
        nN = len(targets)
        x = np.array(x, dtype=np.float32)
        draws = np.ascontiguousarray(draws, dtype=np.float32)
        assert draws.size >= iterations * nN * 2
        bits = np.empty(iterations * nN, dtype=np.uint8)
        trace = np.empty(iterations * nN, dtype=np.float32)
        mm = C.c_float()
        self.lib.orc_warmup_thread_trace(C.c_int(nN), self._targets(targets), _ptr(x, C.c_float), C.c_float(step),
                                         C.c_int(iterations), _ptr(draws, C.c_float), _ptr(bits, C.c_uint8),
                                         _ptr(trace, C.c_float), C.byref(mm))
        return x, bits.reshape(iterations, nN), trace.reshape(iterations, nN), mm.value

    def integrate_r2_thread(self, targets, x, step, samples, draws):
        nN = len(targets)
        x = np.array(x, dtype=np.float32)
        draws = np.ascontiguousarray(draws, dtype=np.float32)
        assert draws.size >= samples * nN * 2
        out = np.empty(nN, dtype=np.float32)
        self.lib.orc_integrate_r2_thread(C.c_int(nN), self._targets(targets), _ptr(x, C.c_float), C.c_float(step),
                                         C.c_int(samples), _ptr(draws, C.c_float), _ptr(out, C.c_float))
        return x, out

    def integrate_sab_thread(self, nA, nB, targets, x, step, samples, b, draws, sigmaNN=4.3, beta=0.2, dmax2=None):
        nN = nA + nB
        assert len(targets) == nN
        x = np.array(x, dtype=np.float32)
        draws = np.ascontiguousarray(draws, dtype=np.float32)
        assert draws.size >= samples * nN * 4
        hits = np.zeros(3, dtype=np.uint64) if dmax2 is not None else None
        ylast = C.c_float()
        mean = self.lib.orc_integrate_sab_thread(C.c_int(nA), C.c_int(nB), self._targets(targets), _ptr(x, C.c_float),
                                                 C.c_float(step), C.c_int(samples), C.c_float(b), C.c_float(sigmaNN),
                                                 C.c_float(beta), _ptr(draws, C.c_float),
                                                 C.c_float(dmax2 if dmax2 is not None else 0.0),
                                                 _ptr(hits, C.c_uint64), C.byref(ylast))
        return x, mean, hits, ylast.value

    def run_estimator_f32(self, run_means):
        r = np.ascontiguousarray(run_means, dtype=np.float32)
        m, se, rms, rmse = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        self.lib.orc_run_estimator_f32(_ptr(r, C.c_float), C.c_int(len(r)), C.byref(m), C.byref(se), C.byref(rms),
                                       C.byref(rmse))
        return m.value, se.value, rms.value, rmse.value

    def cross_sections(self, b, S, Serr):
        b = np.ascontiguousarray(b, dtype=np.float64)
        S = np.ascontiguousarray(S, dtype=np.float64)
        Serr = np.ascontiguousarray(Serr, dtype=np.float64)
        out = np.empty(6, dtype=np.float64)
        self.lib.orc_cross_sections(_ptr(b, C.c_double), _ptr(S, C.c_double), _ptr(Serr, C.c_double), C.c_int(len(b)),
                                    _ptr(out, C.c_double))
        return out

    def target_f32(self, kind, p0, p1, r):
        return self.lib.orc_target_f32(C.c_int(kind), C.c_float(p0), C.c_float(p1), C.c_float(r))

    def init_from_pdf(self, kind, p0, p1, seed, chain, coord):
        self.lib.orc_init_from_pdf.restype = C.c_float
        return self.lib.orc_init_from_pdf(C.c_int(kind), C.c_float(p0), C.c_float(p1), C.c_uint64(seed), C.c_uint64(chain),
                                          C.c_uint32(coord))

    def gamma_nn_r2(self, r, sigmaNN=4.3, beta=0.2):
        return self.lib.orc_GammaNNR2(C.c_float(r), C.c_float(sigmaNN), C.c_float(beta))

    def mvn_chain(self, q_diag, q_off, x, step, sweeps, draws=None, seed=0, chain=0, sweep_offset=0):
        """-> accepted, final x, trace [sweeps][dim], bits [sweeps][dim], min |log U - delta|."""
        dim = len(q_diag)
        qd = np.ascontiguousarray(q_diag, dtype=np.float32)
        qo = np.ascontiguousarray(np.concatenate([q_off, [0.0]]), dtype=np.float32)
        x = np.array(x, dtype=np.float32)
        if draws is not None:
            draws = np.ascontiguousarray(draws, dtype=np.float32)
            assert draws.size >= sweeps * dim * 2
        trace = np.empty((sweeps, dim), dtype=np.float32)
        bits = np.empty((sweeps, dim), dtype=np.uint8)
        mm = C.c_float()
        acc = self.lib.orc_mvn_chain(C.c_int(dim), _ptr(qd, C.c_float), _ptr(qo, C.c_float), _ptr(x, C.c_float),
                                     C.c_float(step), C.c_uint64(sweeps), _ptr(draws, C.c_float), C.c_uint64(seed),
                                     C.c_uint64(chain), C.c_uint64(sweep_offset), _ptr(trace, C.c_float),
                                     _ptr(bits, C.c_uint8), C.byref(mm))
        return acc, x, trace, bits, mm.value

    def baseline_chains_f32(self, chain0, n_chains, nC, kind, p0, p1, step, seed, offset, sweeps, x0=1.0):
        s1, s2 = C.c_double(), C.c_double()
        acc = self.lib.orc_baseline_chains_f32(C.c_uint64(chain0), C.c_uint64(n_chains), C.c_int(nC), C.c_int(kind),
                                               C.c_float(p0), C.c_float(p1), C.c_float(step), C.c_uint64(seed),
                                               C.c_uint64(offset), C.c_uint64(sweeps), C.c_float(x0), C.byref(s1),
                                               C.byref(s2))
        return acc, s1.value, s2.value


class Reference:
    """The reference's own stepsize headers (only available in the build container)."""

    def __init__(self):
        self.lib = load_ref()

    def mh(self, iterations, step, normals, uniforms):
        normals = np.ascontiguousarray(normals, dtype=np.float64)
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
        samples = np.empty(iterations, dtype=np.float64)
        used = C.c_int()
        ratio = self.lib.ref_MetropolisHastingsAlgorithm(C.c_int(iterations), C.c_double(step),
                                                         _ptr(normals, C.c_double), _ptr(uniforms, C.c_double),
                                                         _ptr(samples, C.c_double), C.byref(used))
        return ratio, samples, used.value

    def average(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return self.lib.ref_Average(_ptr(x, C.c_double), C.c_int(len(x)))

    def variance(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return self.lib.ref_Variance(_ptr(x, C.c_double), C.c_int(len(x)))

    def autocorrelation(self, x, lag):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return self.lib.ref_AutoCorrelation(_ptr(x, C.c_double), C.c_int(len(x)), C.c_int(lag))

    def r2ws(self, r, c, a):
        return self.lib.ref_R2WoodsSaxon(C.c_double(r), C.c_double(c), C.c_double(a))

    def r2gauss(self, r, v):
        return self.lib.ref_R2Gaussian(C.c_double(r), C.c_double(v))
