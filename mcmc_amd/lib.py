"""ctypes binding of libmcmc_hip.so (the C ABI declared in include/mcmc_hip.h).

PyTorch is plumbing only: it owns device buffers (tensor.data_ptr()) and the
stream the kernels are launched on.  There is no CPU fallback: if the HIP
library is missing, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCMC_HIP_LIB: another build of the same library (A/B timing of kernel variants, tools/); default: the in-tree build
LIB_PATH = os.environ.get("MCMC_HIP_LIB") or os.path.join(_HERE, "libmcmc_hip.so")

TARGET_WS, TARGET_R2WS, TARGET_GAUSS, TARGET_R2GAUSS = 0, 1, 2, 3
ACCEPT_RECIP, ACCEPT_SHORTCIRCUIT, ACCEPT_NODIV, ACCEPT_LOG, ACCEPT_CROSS = 0, 1, 2, 3, 4
MATH_PRECISE, MATH_FAST = 0, 1
MAX_LAGS = 49
CHAIN_NOALLOC_MAX_BYTES = 8 << 20   # MCMC_CHAIN_NOALLOC_MAX_BYTES
REDUCE_AUTO, REDUCE_HOST, REDUCE_RCCL = -1, 0, 1


def stats_len(n_lags):
    return 3 + 3 * n_lags


class Target(C.Structure):
    _fields_ = [("kind", C.c_int32), ("p0", C.c_float), ("p1", C.c_float)]


class ChainArgs(C.Structure):
    _fields_ = [
        ("n_chains", C.c_uint64), ("n_coords", C.c_uint32), ("n_lags", C.c_uint32), ("n_steps", C.c_uint64),
        ("step_size", C.c_double), ("center", C.c_double), ("target", Target),
        ("precision", C.c_int32), ("accept_rule", C.c_int32), ("math_mode", C.c_int32), ("window_start", C.c_int32),
        ("seed", C.c_uint64), ("step_offset", C.c_uint64), ("chain_offset", C.c_uint64),
        ("state", C.c_void_p), ("stats", C.c_void_p), ("ring", C.c_void_p), ("samples", C.c_void_p),
        ("inj_normals", C.c_void_p), ("inj_uniforms", C.c_void_p), ("accept_bits", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class ChainSummary(C.Structure):
    _fields_ = [("mean", C.c_double), ("variance", C.c_double), ("acceptance", C.c_double), ("tau_int", C.c_double)]


class MvnArgs(C.Structure):
    _fields_ = [
        ("n_chains", C.c_uint64), ("dim", C.c_uint32), ("observable", C.c_uint32), ("n_steps", C.c_uint64),
        ("step_size", C.c_float), ("center", C.c_float), ("q_diag", C.c_void_p), ("q_off", C.c_void_p),
        ("math_mode", C.c_int32), ("lag", C.c_uint32),
        ("seed", C.c_uint64), ("step_offset", C.c_uint64), ("chain_offset", C.c_uint64),
        ("window_begin", C.c_uint64), ("window_end", C.c_uint64),
        ("state", C.c_void_p), ("ring", C.c_void_p), ("samples_obs", C.c_void_p), ("samples_full", C.c_void_p),
        ("inj", C.c_void_p), ("accept_bits", C.c_void_p), ("stats", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class XsArgs(C.Structure):
    _fields_ = [
        ("nNucleonsA", C.c_uint32), ("nNucleonsB", C.c_uint32), ("totalThreads", C.c_uint64),
        ("targets", C.c_void_p), ("prevSample", C.c_void_p), ("stepsize", C.c_float),
        ("accept_rule", C.c_int32), ("math_mode", C.c_int32), ("uniform_kind", C.c_int32),
        ("seed", C.c_uint64), ("step_offset", C.c_uint64), ("event_offset", C.c_uint64),
        ("injected", C.c_void_p), ("injected_stride", C.c_uint64),
        ("sigmaNN", C.c_float), ("beta", C.c_float),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class Config(C.Structure):
    _fields_ = [
        ("nNeutronA", C.c_int32), ("nProtonA", C.c_int32), ("nNeutronB", C.c_int32), ("nProtonB", C.c_int32),
        ("neutronA", Target), ("protonA", Target), ("neutronB", Target), ("protonB", Target),
        ("totalBlocks", C.c_int32), ("threadsPerBlock", C.c_int32), ("samplesPerThread", C.c_int32),
        ("totalRuns", C.c_int32), ("samplesForMix", C.c_int32),
        ("b0", C.c_float), ("db", C.c_float), ("ndbs", C.c_int32),
        ("seed", C.c_uint64), ("stepsize", C.c_float), ("initSample", C.c_float),
        ("sigmaNN", C.c_float), ("beta", C.c_float),
        ("accept_rule", C.c_int32), ("math_mode", C.c_int32), ("device", C.c_int32),
        ("event_offset", C.c_uint64),
        ("integrateRMS", C.c_int32), ("integrateSAB", C.c_int32), ("outDir", C.c_char_p), ("outFile", C.c_char_p),
        ("nucleon_targets", C.POINTER(Target)),
    ]


# Every symbol include/mcmc_hip.h declares (tests/test_abi.py checks the header against this list).
EXPORTS = [
    "mcmc_chain_workspace_bytes", "mcmc_chain_run", "mcmc_chain_run_noalloc", "mcmc_chain_clock_probe",
    "mcmc_stats_finalize", "mcmc_launch_draws",
    "mcmc_mvn_workspace_bytes", "mcmc_mvn_run", "mcmc_mvn_finalize",
    "mcmc_xs_workspace_bytes", "mcmc_launch_init_sample_array", "mcmc_launch_warmup_metropolis",
    "mcmc_launch_init_sample_from_pdf", "mcmc_launch_integrate_r2", "mcmc_launch_integrate_sab", "mcmc_launch_integrate_inel", "mcmc_launch_row_mean_f32",
    "mcmc_row_mean_workspace_bytes",
    "mcmc_Sum", "mcmc_Average", "mcmc_Variance", "mcmc_AutoCorrelation", "mcmc_Variance_n", "mcmc_AutoCorrelation_n",
    "mcmc_Average_f64", "mcmc_Variance_f64", "mcmc_AutoCorrelation_f64", "mcmc_cross_sections",
    "mcmc_config_default", "mcmc_create", "mcmc_destroy", "mcmc_warmup", "mcmc_integrate_r2",
    "mcmc_integrate_sab", "mcmc_run_rms", "mcmc_run_sab_scan", "mcmc_run_all", "mcmc_set_option", "mcmc_run_inel_scan", "mcmc_init_from_pdf", "mcmc_get_state", "mcmc_wait",
    "mcmc_row_sum_fixed_workspace_bytes", "mcmc_launch_row_sum_fixed_f32", "mcmc_fixed_sum_value",
    "mcmc_multi_create", "mcmc_multi_destroy", "mcmc_multi_info", "mcmc_multi_shard", "mcmc_multi_warmup",
    "mcmc_multi_init_from_pdf", "mcmc_multi_run_rms", "mcmc_multi_run_sab_scan", "mcmc_multi_run_inel_scan",
    "mcmc_stepsize_run_multi",
    "mcmc_stepsize_run", "mcmc_tune_stepsize", "mcmc_isotope", "mcmc_last_error", "mcmc_version", "mcmc_philox_rounds", "mcmc_device_count",
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()); "
            "mcmc_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.mcmc_chain_workspace_bytes.restype = C.c_size_t
    lib.mcmc_chain_workspace_bytes.argtypes = [P(ChainArgs)]
    lib.mcmc_chain_run.argtypes = [P(ChainArgs), C.c_void_p]
    lib.mcmc_chain_run_noalloc.argtypes = [P(ChainArgs), C.c_void_p]
    lib.mcmc_chain_clock_probe.argtypes = [P(ChainArgs), P(C.c_double), C.c_void_p]
    lib.mcmc_stats_finalize.argtypes = [P(C.c_double), C.c_uint32, C.c_uint64, C.c_uint64, C.c_double,
                                        P(ChainSummary), P(C.c_double)]
    lib.mcmc_launch_draws.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
    lib.mcmc_mvn_workspace_bytes.restype = C.c_size_t
    lib.mcmc_mvn_workspace_bytes.argtypes = [P(MvnArgs)]
    lib.mcmc_mvn_run.argtypes = [P(MvnArgs), C.c_void_p]
    lib.mcmc_mvn_finalize.argtypes = [P(C.c_double), C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_double,
                                      P(ChainSummary), P(C.c_double)]
    lib.mcmc_xs_workspace_bytes.restype = C.c_size_t
    lib.mcmc_xs_workspace_bytes.argtypes = [P(XsArgs)]
    lib.mcmc_launch_init_sample_array.argtypes = [C.c_void_p, C.c_float, C.c_uint32, C.c_uint64, C.c_void_p]
    lib.mcmc_launch_warmup_metropolis.argtypes = [P(XsArgs), C.c_uint32, C.c_void_p]
    lib.mcmc_launch_integrate_r2.argtypes = [P(XsArgs), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.mcmc_launch_integrate_sab.argtypes = [P(XsArgs), C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p,
                                              C.c_float, C.c_void_p]
    lib.mcmc_launch_init_sample_from_pdf.argtypes = [P(XsArgs), C.c_void_p]
    lib.mcmc_launch_integrate_inel.argtypes = [P(XsArgs), C.c_uint32, C.c_float, C.c_void_p, C.c_float, C.c_void_p]
    lib.mcmc_launch_row_mean_f32.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t,
                                             C.c_void_p]
    lib.mcmc_row_mean_workspace_bytes.restype = C.c_size_t
    lib.mcmc_row_mean_workspace_bytes.argtypes = [C.c_uint32, C.c_uint64]
    for f in ("mcmc_Sum", "mcmc_Average"):
        getattr(lib, f).restype = C.c_float
        getattr(lib, f).argtypes = [P(C.c_float), C.c_int]
    lib.mcmc_Variance.restype = C.c_float
    lib.mcmc_Variance.argtypes = [P(C.c_float), C.c_float, C.c_int]
    lib.mcmc_AutoCorrelation.restype = C.c_float
    lib.mcmc_AutoCorrelation.argtypes = [P(C.c_float), C.c_float, C.c_int, C.c_int]
    lib.mcmc_Variance_n.restype = C.c_float
    lib.mcmc_Variance_n.argtypes = [P(C.c_float), C.c_int]
    lib.mcmc_AutoCorrelation_n.restype = C.c_float
    lib.mcmc_AutoCorrelation_n.argtypes = [P(C.c_float), C.c_int, C.c_int]
    for f in ("mcmc_Average_f64", "mcmc_Variance_f64"):
        getattr(lib, f).restype = C.c_double
        getattr(lib, f).argtypes = [P(C.c_double), C.c_int]
    lib.mcmc_AutoCorrelation_f64.restype = C.c_double
    lib.mcmc_AutoCorrelation_f64.argtypes = [P(C.c_double), C.c_int, C.c_int]
    lib.mcmc_cross_sections.argtypes = [P(C.c_double), P(C.c_double), P(C.c_double), C.c_int, P(C.c_double)]
    lib.mcmc_config_default.restype = None
    lib.mcmc_config_default.argtypes = [P(Config)]
    lib.mcmc_create.argtypes = [P(Config), P(C.c_void_p)]
    lib.mcmc_destroy.argtypes = [C.c_void_p]
    lib.mcmc_warmup.argtypes = [C.c_void_p, C.c_float, C.c_int]
    lib.mcmc_integrate_r2.argtypes = [C.c_void_p, C.c_float, C.c_int, P(C.c_float)]
    lib.mcmc_integrate_sab.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_float, P(C.c_float), P(C.c_double),
                                       C.c_float]
    lib.mcmc_run_rms.argtypes = [C.c_void_p, P(C.c_float), P(C.c_float)]
    lib.mcmc_run_sab_scan.argtypes = [C.c_void_p, P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_double), C.c_float]
    lib.mcmc_run_all.argtypes = [C.c_void_p, P(C.c_double)]
    lib.mcmc_set_option.argtypes = [C.c_char_p, C.c_char_p]
    lib.mcmc_init_from_pdf.argtypes = [C.c_void_p]
    lib.mcmc_run_inel_scan.argtypes = [C.c_void_p, P(C.c_float), P(C.c_double), P(C.c_double), C.c_float]
    lib.mcmc_get_state.argtypes = [C.c_void_p, P(C.c_float)]
    lib.mcmc_wait.argtypes = [C.c_void_p]
    lib.mcmc_stepsize_run.argtypes = [P(Target), C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint32,
                                      C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_double, P(ChainSummary),
                                      P(C.c_double), C.c_void_p]
    lib.mcmc_row_sum_fixed_workspace_bytes.restype = C.c_size_t
    lib.mcmc_row_sum_fixed_workspace_bytes.argtypes = [C.c_uint32]
    lib.mcmc_launch_row_sum_fixed_f32.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t,
                                                  C.c_void_p]
    lib.mcmc_fixed_sum_value.restype = C.c_double
    lib.mcmc_fixed_sum_value.argtypes = [P(C.c_double)]
    lib.mcmc_multi_create.argtypes = [P(Config), P(C.c_int), C.c_int, C.c_int, P(C.c_void_p)]
    lib.mcmc_multi_destroy.argtypes = [C.c_void_p]
    lib.mcmc_multi_info.argtypes = [C.c_void_p, P(C.c_int), P(C.c_int), P(C.c_uint64), P(C.c_uint64)]
    lib.mcmc_multi_shard.restype = C.c_void_p
    lib.mcmc_multi_shard.argtypes = [C.c_void_p, C.c_int]
    lib.mcmc_multi_warmup.argtypes = [C.c_void_p, C.c_float, C.c_int]
    lib.mcmc_multi_init_from_pdf.argtypes = [C.c_void_p]
    lib.mcmc_multi_run_rms.argtypes = [C.c_void_p, P(C.c_float), P(C.c_float)]
    lib.mcmc_multi_run_sab_scan.argtypes = [C.c_void_p, P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_double), C.c_float]
    lib.mcmc_multi_run_inel_scan.argtypes = [C.c_void_p, P(C.c_float), P(C.c_double), P(C.c_double), C.c_float]
    lib.mcmc_stepsize_run_multi.argtypes = [P(Target), C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint32,
                                            C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_double, P(C.c_int), C.c_int,
                                            C.c_int, P(ChainSummary), P(C.c_double), C.c_void_p]
    lib.mcmc_isotope.argtypes = [C.c_char_p, P(C.c_int), P(C.c_int)]
    lib.mcmc_tune_stepsize.argtypes = [P(Target), C.c_double, C.c_double, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64,
                                       C.c_uint64, C.c_double, P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_double)]
    lib.mcmc_last_error.restype = C.c_char_p
    lib.mcmc_version.restype = C.c_char_p
    lib.mcmc_device_count.restype = C.c_int
    lib.mcmc_philox_rounds.restype = C.c_int
    return lib


lib = _load()


class McmcError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise McmcError(f"mcmc_hip error {rc}: {lib.mcmc_last_error().decode()}")
