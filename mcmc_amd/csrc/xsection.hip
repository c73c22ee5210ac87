// xsection.hip -- Layer A of the xsection path: argument checks, kernel
// selection and launches for InitSampleArray / WarmupMetropolis /
// MCIntegrate_R2 / MCIntegrate_S_AB (xsection/src/main.cu:130-149,197,335) and
// the device row means that replace the host Average of main.cu:221-223,359.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "host_common.hpp"
#include "xs_kernel.hpp"

using namespace mcmc;

namespace mcmc {
// one translation unit per MODE of sab_kernel (sab_mode.hip, -DMCMC_SAB_MODE=1..4)
int launch_sab_mode_1(const mcmc_xs_args *, const XsParams &, int G, size_t lds, bool all_r2ws, hipStream_t);
int launch_sab_mode_2(const mcmc_xs_args *, const XsParams &, int G, size_t lds, bool all_r2ws, hipStream_t);
int launch_sab_mode_3(const mcmc_xs_args *, const XsParams &, int G, size_t lds, bool all_r2ws, hipStream_t);
int launch_sab_mode_4(const mcmc_xs_args *, const XsParams &, int G, size_t lds, bool all_r2ws, hipStream_t);
} // namespace mcmc

namespace mcmc {
extern int g_mvn_reg;  // mvn.hip
extern int g_lag_mfma;      // chain.hip
extern int g_samples_pairs; // chain.hip
}

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Diagnostic switches: read from the environment ONCE, at the first launch that looks at them;
// mcmc_set_option overrides them afterwards (tests and tools compare kernel variants inside one process).
struct SabOptions {
    int cull = -1;   // MCMC_SAB_CULL:  0 = no pair culling, otherwise on
    int small = -1;  // MCMC_SAB_SMALL: 0 / 1 force the general / the register kernel, -1 = by applicability
    int plan_g = 0, plan_gl = 0; // MCMC_SAB_PLAN="G,ga_log2" (0 = cost model)
};
std::mutex opt_mutex;
SabOptions opt_state;
bool opt_loaded = false;

void parse_option(SabOptions &o, const char *name, const char *v)
{
    if (std::strcmp(name, "MCMC_SAB_CULL") == 0) o.cull = v && v[0] == '0' ? 0 : -1;
    else if (std::strcmp(name, "MCMC_SAB_SMALL") == 0) o.small = v && v[0] == '0' ? 0 : v && v[0] == '1' ? 1 : -1;
    else if (std::strcmp(name, "MCMC_SAB_PLAN") == 0) {
        int g = 0, gl = 0;
        o.plan_g = o.plan_gl = 0;
        if (v && std::sscanf(v, "%d,%d", &g, &gl) == 2 && (g == 4 || g == 16 || g == 64) && gl >= 0 && (1 << gl) <= g)
            o.plan_g = g, o.plan_gl = gl;
    }
}

SabOptions sab_options()
{
    std::lock_guard<std::mutex> lock(opt_mutex);
    if (!opt_loaded) {
        for (const char *name : {"MCMC_SAB_CULL", "MCMC_SAB_SMALL", "MCMC_SAB_PLAN"})
            if (const char *v = std::getenv(name)) parse_option(opt_state, name, v);
        opt_loaded = true;
    }
    return opt_state;
}

size_t rowmean_bytes(uint32_t rows) { return align_up((size_t)rows * RR_CHUNKS * sizeof(double), 256); }

int check_xs(const mcmc_xs_args *a)
{
    if (!a) return set_error(MCMC_EINVAL, "null args");
    if (a->nNucleonsA + a->nNucleonsB == 0) return set_error(MCMC_EINVAL, "no nucleons");
    if (a->nNucleonsA + a->nNucleonsB > 65535) return set_error(MCMC_EINVAL, "nNucleons must be <= 65535");
    if (a->totalThreads == 0) return set_error(MCMC_EINVAL, "totalThreads must be > 0");
    // grids: ceil(T / 4) blocks at one event per wavefront must fit 2^31 - 1
    if (a->totalThreads > (1ull << 32)) return set_error(MCMC_EINVAL, "totalThreads must be <= 2^32 per device");
    // the Philox counter carries 40 bits of the global event id (device_common.hpp draw_block)
    if (a->event_offset > (1ull << 40) || a->event_offset + a->totalThreads > (1ull << 40))
        return set_error(MCMC_EINVAL, "event_offset + totalThreads must be <= 2^40 (global event ids are 40-bit Philox counters)");
    if (!a->targets || !a->prevSample) return set_error(MCMC_EINVAL, "targets and prevSample must be device pointers");
    if (a->accept_rule != MCMC_ACCEPT_RECIP && a->accept_rule != MCMC_ACCEPT_NODIV && a->accept_rule != MCMC_ACCEPT_CROSS &&
        a->accept_rule != MCMC_ACCEPT_LOG)
        return set_error(MCMC_EINVAL, "xsection kernels implement MCMC_ACCEPT_RECIP, _NODIV, _CROSS and _LOG");
    if (a->math_mode == MCMC_MATH_PRECISE && a->accept_rule != MCMC_ACCEPT_RECIP)
        return set_error(MCMC_EINVAL, "precise math implements MCMC_ACCEPT_RECIP only");
    if (a->math_mode != MCMC_MATH_PRECISE && a->math_mode != MCMC_MATH_FAST) return set_error(MCMC_EINVAL, "bad math_mode");
    return MCMC_OK;
}

XsParams make_params(const mcmc_xs_args *a, uint32_t samples)
{
    XsParams p{};
    p.T = a->totalThreads;
    p.seed = a->seed;
    p.step_offset = a->step_offset;
    p.event_offset = a->event_offset;
    p.inj_stride = a->injected_stride;
    p.nA = a->nNucleonsA;
    p.nB = a->nNucleonsB;
    p.samples = samples;
    p.stepsize = a->stepsize;
    p.targets = a->targets;
    p.prevSample = a->prevSample;
    p.injected = a->injected;
    // IntegrateSAB.cuh:10-12 in the reference's float arithmetic
    const float PI = 3.141592653589f; // Common.cuh:21
    p.G0 = a->sigmaNN / (4 * PI * a->beta);
    p.twoBeta = 2.f * a->beta;
    p.pos_scale = 1.0f;
    return p;
}

template <bool R2>
int launch_chain_xs(const mcmc_xs_args *a, const XsParams &p, hipStream_t s)
{
    const dim3 grid((unsigned)((p.T + XS_BLOCK - 1) / XS_BLOCK), p.nA + p.nB);
    const dim3 block(XS_BLOCK);
    const bool inj = a->injected != nullptr;
    if (a->math_mode == MCMC_MATH_PRECISE) {
        if (inj) hipLaunchKernelGGL((xs_chain_kernel<MCMC_ACCEPT_RECIP, MCMC_MATH_PRECISE, true, R2>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((xs_chain_kernel<MCMC_ACCEPT_RECIP, MCMC_MATH_PRECISE, false, R2>), grid, block, 0, s, p);
    } else if (a->accept_rule == MCMC_ACCEPT_RECIP) {
        if (inj) hipLaunchKernelGGL((xs_chain_kernel<MCMC_ACCEPT_RECIP, MCMC_MATH_FAST, true, R2>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((xs_chain_kernel<MCMC_ACCEPT_RECIP, MCMC_MATH_FAST, false, R2>), grid, block, 0, s, p);
    } else if (a->accept_rule == MCMC_ACCEPT_CROSS) {
        if (inj) hipLaunchKernelGGL((xs_chain_kernel<MCMC_ACCEPT_CROSS, MCMC_MATH_FAST, true, R2>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((xs_chain_kernel<MCMC_ACCEPT_CROSS, MCMC_MATH_FAST, false, R2>), grid, block, 0, s, p);
    } else if (a->accept_rule == MCMC_ACCEPT_LOG) {
        if (inj) hipLaunchKernelGGL((xs_chain_kernel<MCMC_ACCEPT_LOG, MCMC_MATH_FAST, true, R2>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((xs_chain_kernel<MCMC_ACCEPT_LOG, MCMC_MATH_FAST, false, R2>), grid, block, 0, s, p);
    } else {
        if (inj) hipLaunchKernelGGL((xs_chain_kernel<MCMC_ACCEPT_NODIV, MCMC_MATH_FAST, true, R2>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((xs_chain_kernel<MCMC_ACCEPT_NODIV, MCMC_MATH_FAST, false, R2>), grid, block, 0, s, p);
    }
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

template <typename T>
int row_reduce(const T *x, uint32_t rows, uint64_t cols, double scale, double *out, int accumulate, void *ws, size_t ws_bytes,
               hipStream_t s)
{
    if (ws == nullptr || ws_bytes < rowmean_bytes(rows)) return set_error(MCMC_ENOMEM, "row reduction workspace too small");
    hipLaunchKernelGGL((row_partial_kernel<T>), dim3(RR_CHUNKS, rows), dim3(256), 0, s, x, cols, (double *)ws);
    MCMC_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(row_final_kernel, dim3((rows + 63) / 64), dim3(64), 0, s, (const double *)ws, out, rows, scale,
                       accumulate);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

// Lanes per event G in {4,16,64} and their GA x GB split over the A x B pair grid: the
// combination with the least padded work that still leaves >= 2 blocks per CU.
struct SabPlan {
    int G;
    int ga_log2;
    size_t lds;
};

size_t sab_lds_bytes(int G, uint32_t nA, uint32_t nB, bool hits)
{
    const size_t nN = (size_t)nA + nB;
    const size_t epb = XS_BLOCK / G;
    // participant marks are bytes: with uint32 marks four Pb-Pb events took 54.9 KB and only two blocks fitted a CU
    return xs_sab_header_bytes(nN, hits) + 3 * sizeof(float2) * epb * nN + (hits ? align_up(epb * nN, 16) : 0);
}

SabPlan sab_plan(uint32_t nA, uint32_t nB, bool hits, const SabOptions &opt)
{
    // tuning hook: MCMC_SAB_PLAN="G,ga_log2" overrides the cost model (tools/sab_plan_sweep.py)
    if (opt.plan_g != 0) return SabPlan{opt.plan_g, opt.plan_gl, sab_lds_bytes(opt.plan_g, nA, nB, hits)};
    const int cand[3] = {4, 16, 64};
    SabPlan best{64, 0, sab_lds_bytes(64, nA, nB, hits)};
    double best_cost = 1e300;
    for (int G : cand) {
        const size_t lds = sab_lds_bytes(G, nA, nB, hits);
        if (lds > 64 * 1024 && G != 64) continue; // keep >= 2 blocks per CU unless there is no choice
        const double nN = (double)nA + nB;
        const double mh = (double)((uint32_t)(nN + G - 1) / G) * 160.0;
        for (int gl = 0; (1 << gl) <= G; ++gl) {
            const uint32_t GA = 1u << gl, GB = (uint32_t)G >> gl;
            const uint32_t ia = (nA + GA - 1) / GA, jb = (nB + GB - 1) / GB;
            // pair phase per lane: every pass over the lane's A nucleons holds JC in {16,8,4,2,1} B nucleons;
            // one iteration costs ~10 issue slots per pair + ~14 of loop / LDS overhead
            double pair = 0.0;
            uint32_t rem = jb;
            for (uint32_t jc = 16; jc >= 1; jc >>= 1)
                while (rem >= jc) {
                    pair += (double)ia * (jc * 10.0 + 14.0) + 30.0;
                    rem -= jc;
                }
            const double cost = G * (mh + pair + 12.0 * __builtin_ctz(G));
            if (cost < best_cost) {
                best_cost = cost;
                best = SabPlan{G, gl, lds};
            }
        }
    }
    return best;
}

int launch_sab(int mode, const mcmc_xs_args *a, const XsParams &p, const SabPlan &plan, bool all_r2ws, hipStream_t s)
{
    switch (mode) {
    case 1: return launch_sab_mode_1(a, p, plan.G, plan.lds, all_r2ws, s);
    case 2: return launch_sab_mode_2(a, p, plan.G, plan.lds, all_r2ws, s);
    case 3: return launch_sab_mode_3(a, p, plan.G, plan.lds, all_r2ws, s);
    default: return launch_sab_mode_4(a, p, plan.G, plan.lds, all_r2ws, s);
    }
}

// Distance, in the units of the stored positions, beyond which a nucleon pair changes no result (sab_cull in
// xs_kernel.hpp).  Product: 1 - Gamma rounds to exactly 1.0f once |Gamma| <= 2^-25, i.e. once
// d^2 / (2 beta) * log2(e) >= 25 + log2|G0|; 0.05 on that exponent (3.5 % on Gamma) covers the error of
// v_exp_f32 / expf and of the coordinate differences by orders of magnitude.  Hits: d^2 < dmax2 is impossible
// beyond sqrt(dmax2) (0.1 % of margin).  MCMC_SAB_CULL=0 switches the culling off (tests compare the two).
float sab_cull_distance(const mcmc_xs_args *a, const XsParams &p, bool prod, bool hit, const SabOptions &opt)
{
    if (opt.cull == 0) return 0.0f;
    double c2 = 0.0;
    if (prod) {
        if (!(std::fabs((double)p.G0) > 0.0) || !std::isfinite((double)p.G0)) return 0.0f;
        double sexp = 25.05 + std::log2(std::fabs((double)p.G0));
        if (sexp < 0.0) sexp = 0.0;
        c2 = a->math_mode == MCMC_MATH_FAST ? sexp : sexp * 0.6931471805599453 * (double)p.twoBeta;
    }
    if (hit && (double)p.hit_d2 * 1.001 > c2) c2 = (double)p.hit_d2 * 1.001;
    if (!(c2 > 0.0) || !std::isfinite(c2)) return 0.0f;
    return (float)(std::sqrt(c2) * 1.0005);
}

// Register-resident kernel for light nuclei (sab_small_kernel): NA, NB in 1..4.
template <int NA, int NB>
int launch_sab_small_ab(const mcmc_xs_args *a, const XsParams &p, hipStream_t s)
{
    const dim3 grid((unsigned)((p.T + XS_BLOCK - 1) / XS_BLOCK)), block(XS_BLOCK);
    if (a->accept_rule == MCMC_ACCEPT_RECIP) hipLaunchKernelGGL((sab_small_kernel<NA, NB, MCMC_ACCEPT_RECIP>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((sab_small_kernel<NA, NB, MCMC_ACCEPT_NODIV>), grid, block, 0, s, p);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

template <int NA>
int launch_sab_small_a(const mcmc_xs_args *a, const XsParams &p, hipStream_t s)
{
    switch (p.nB) {
    case 1: return launch_sab_small_ab<NA, 1>(a, p, s);
    case 2: return launch_sab_small_ab<NA, 2>(a, p, s);
    case 3: return launch_sab_small_ab<NA, 3>(a, p, s);
    default: return launch_sab_small_ab<NA, 4>(a, p, s);
    }
}

int launch_sab_small(const mcmc_xs_args *a, const XsParams &p, hipStream_t s)
{
    switch (p.nA) {
    case 1: return launch_sab_small_a<1>(a, p, s);
    case 2: return launch_sab_small_a<2>(a, p, s);
    case 3: return launch_sab_small_a<3>(a, p, s);
    default: return launch_sab_small_a<4>(a, p, s);
    }
}

// product only, Philox draws, FAST math, all densities r^2 Woods-Saxon, at most 4 + 4 nucleons.
// One event per lane needs >= 2 waves per SIMD of events to beat the general kernel's 4-16 lanes per
// event (measured, He-4 + He-4: 1.96e10 vs 2.58e10 event samples/s at 2^15 events, 4.6e10 vs 3.8e10 at
// 2^17, 5.65e10 vs 4.57e10 at 2^21), hence the event-count threshold.
// MCMC_SAB_SMALL=0 / =1 force the general / the register kernel (tests compare the two)
// The two kernels give bit-identical results (the register kernel multiplies in the general kernel's G = 4 order),
// so a choice that depends on this device's share of the events does not show in them.
bool sab_small_applies(const mcmc_xs_args *a, bool product_only, const SabPlan &plan, const SabOptions &opt)
{
    if (!product_only || a->injected || a->math_mode != MCMC_MATH_FAST || a->uniform_kind != MCMC_TARGET_R2WS) return false;
    if (a->accept_rule != MCMC_ACCEPT_RECIP && a->accept_rule != MCMC_ACCEPT_NODIV && a->accept_rule != MCMC_ACCEPT_CROSS)
        return false;
    if (a->nNucleonsA < 1 || a->nNucleonsA > 4 || a->nNucleonsB < 1 || a->nNucleonsB > 4) return false;
    if (plan.G != 4) return false;
    if (opt.small == 0) return false;
    if (opt.small == 1) return true;
    return a->totalThreads >= (1ull << 17);
}

} // namespace

// Diagnostic switches, settable from inside the process (value NULL: back to the default).  No reference counterpart.
extern "C" int mcmc_set_option(const char *name, const char *value)
{
    if (!name) return set_error(MCMC_EINVAL, "null option name");
    if (std::strcmp(name, "MCMC_MVN_REG") == 0) { // stress target: 0 / 1 = LDS / register kernel, NULL = default
        g_mvn_reg = value && value[0] == '0' ? 0 : value && value[0] == '1' ? 1 : -1;
        return MCMC_OK;
    }
    if (std::strcmp(name, "MCMC_LAG_MFMA") == 0) { // FP32 chains with lags: 0 = lag products on the VALU, else the matrix cores
        g_lag_mfma = value && value[0] == '0' ? 0 : 1;
        return MCMC_OK;
    }
    if (std::strcmp(name, "MCMC_SAMPLES_PAIRS") == 0) { // chain sample write-back: 1 = [n_steps / 2][n_coords][n_chains][2]
        g_samples_pairs = value && value[0] == '1' ? 1 : 0;
        return MCMC_OK;
    }
    if (std::strcmp(name, "MCMC_SAB_CULL") && std::strcmp(name, "MCMC_SAB_SMALL") && std::strcmp(name, "MCMC_SAB_PLAN"))
        return set_error(MCMC_EINVAL, "unknown option (MCMC_SAB_CULL, MCMC_SAB_SMALL, MCMC_SAB_PLAN, MCMC_MVN_REG, MCMC_LAG_MFMA, MCMC_SAMPLES_PAIRS)");
    (void)sab_options(); // the environment first, so that this call wins
    std::lock_guard<std::mutex> lock(opt_mutex);
    parse_option(opt_state, name, value);
    return MCMC_OK;
}

extern "C" size_t mcmc_row_mean_workspace_bytes(uint32_t rows, uint64_t) { return rowmean_bytes(rows); }

extern "C" size_t mcmc_xs_workspace_bytes(const mcmc_xs_args *a)
{
    if (!a) return 0;
    const uint32_t nN = a->nNucleonsA + a->nNucleonsB;
    return rowmean_bytes(nN < 4 ? 4 : nN) + align_up(3 * sizeof(uint32_t) * a->totalThreads, 256);
}

extern "C" int mcmc_launch_init_sample_array(float *prevSample, float initSample, uint32_t nNucleons, uint64_t totalThreads,
                                             void *stream)
{
    if (!prevSample) return set_error(MCMC_EINVAL, "prevSample is null");
    const uint64_t total = (uint64_t)nNucleons * totalThreads;
    if (total == 0) return set_error(MCMC_EINVAL, "empty sample array");
    if ((total + 255) / 256 > 0x7fffffffull) return set_error(MCMC_EINVAL, "nNucleons * totalThreads must be < 2^39");
    hipLaunchKernelGGL(init_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       prevSample, initSample, total);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

extern "C" int mcmc_launch_init_sample_from_pdf(const mcmc_xs_args *a, void *stream)
{
    int rc = check_xs(a);
    if (rc) return rc;
    const uint32_t nN = a->nNucleonsA + a->nNucleonsB;
    const uint64_t tiles = (a->totalThreads + XS_BLOCK - 1) / XS_BLOCK;
    const dim3 grid((unsigned)(tiles < 1024 ? tiles : 1024), nN);
    hipLaunchKernelGGL(init_from_pdf_kernel, grid, dim3(XS_BLOCK), 0, (hipStream_t)stream, a->targets, a->prevSample,
                       a->totalThreads, a->seed, a->event_offset);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

extern "C" int mcmc_launch_warmup_metropolis(const mcmc_xs_args *a, uint32_t iterations, void *stream)
{
    int rc = check_xs(a);
    if (rc) return rc;
    if (a->injected && a->injected_stride < (uint64_t)iterations * (a->nNucleonsA + a->nNucleonsB) * 2)
        return set_error(MCMC_EINVAL, "injected_stride too small for iterations * nNucleons * 2 draws");
    const XsParams p = make_params(a, iterations);
    if (iterations == 0) return MCMC_OK;
    return launch_chain_xs<false>(a, p, (hipStream_t)stream);
}

extern "C" int mcmc_launch_integrate_r2(const mcmc_xs_args *a, float *resultBuffer, uint32_t samples, double *runMean,
                                        void *stream)
{
    int rc = check_xs(a);
    if (rc) return rc;
    if (!resultBuffer) return set_error(MCMC_EINVAL, "resultBuffer is null");
    if (samples == 0) return set_error(MCMC_EINVAL, "samples must be > 0");
    const uint32_t nN = a->nNucleonsA + a->nNucleonsB;
    if (a->injected && a->injected_stride < (uint64_t)samples * nN * 2)
        return set_error(MCMC_EINVAL, "injected_stride too small for samples * nNucleons * 2 draws");
    XsParams p = make_params(a, samples);
    p.result = resultBuffer;
    rc = launch_chain_xs<true>(a, p, (hipStream_t)stream);
    if (rc) return rc;
    if (runMean)
        return row_reduce<float>(resultBuffer, nN, a->totalThreads, 1.0 / (double)a->totalThreads, runMean, 0, a->workspace,
                                 a->workspace_bytes, (hipStream_t)stream);
    return MCMC_OK;
}

extern "C" int mcmc_launch_integrate_sab(const mcmc_xs_args *a, float *resultBuffer, uint32_t samples, float impactParameter,
                                         double *runMean, double *hits, float dmax2, void *stream)
{
    int rc = check_xs(a);
    if (rc) return rc;
    if (!resultBuffer && !hits) return set_error(MCMC_EINVAL, "resultBuffer and hits are both null: nothing to compute");
    if (!resultBuffer && runMean) return set_error(MCMC_EINVAL, "runMean needs resultBuffer");
    if (samples == 0) return set_error(MCMC_EINVAL, "samples must be > 0");
    if (a->nNucleonsA == 0 || a->nNucleonsB == 0) return set_error(MCMC_EINVAL, "both nuclei need at least one nucleon");
    const uint32_t nN = a->nNucleonsA + a->nNucleonsB;
    if (a->injected && a->injected_stride < (uint64_t)samples * nN * 4)
        return set_error(MCMC_EINVAL, "injected_stride too small for samples * nNucleons * 4 draws");
    if (!(a->beta > 0.0f)) return set_error(MCMC_EINVAL, "beta must be > 0");
    if ((runMean || hits) && (!a->workspace || a->workspace_bytes < mcmc_xs_workspace_bytes(a)))
        return set_error(MCMC_ENOMEM, "workspace smaller than mcmc_xs_workspace_bytes()");
    hipStream_t s = (hipStream_t)stream;

    XsParams p = make_params(a, samples);
    p.result = resultBuffer;
    // FAST: positions are stored in units of sqrt(2 beta / log2 e), so Gamma = G0 * 2^(-d'^2)
    if (a->math_mode == MCMC_MATH_FAST) p.pos_scale = sqrtf(1.4426950408889634f / (2.0f * a->beta));
    p.b_stored = impactParameter * p.pos_scale;
    p.hit_d2 = dmax2 * p.pos_scale * p.pos_scale;
    uint32_t *hitbuf = nullptr;
    if (hits) {
        hitbuf = (uint32_t *)((char *)a->workspace + rowmean_bytes(nN < 4 ? 4 : nN));
        p.hits = hitbuf;
    }

    // `targets` is a device array, so the caller says whether one kind covers all nucleons.
    const bool all_r2ws = a->uniform_kind == MCMC_TARGET_R2WS;

    const SabOptions opt = sab_options();
    const SabPlan plan = sab_plan(a->nNucleonsA, a->nNucleonsB, hits != nullptr, opt);
    if (plan.lds > 160 * 1024) return set_error(MCMC_EINVAL, "nuclei too large for the LDS-resident S_AB kernel (A+B <= ~1600)");
    p.ga_log2 = (uint32_t)plan.ga_log2;
    p.cull_d = sab_cull_distance(a, p, resultBuffer != nullptr, hits != nullptr, opt);
    if (sab_small_applies(a, resultBuffer && !hits, plan, opt)) rc = launch_sab_small(a, p, s);
    else rc = launch_sab(resultBuffer && hits ? 3 : resultBuffer ? 1 : 2, a, p, plan, all_r2ws, s);
    if (rc) return rc;
    if (runMean) {
        rc = row_reduce<float>(resultBuffer, 1, a->totalThreads, 1.0 / (double)a->totalThreads, runMean, 0, a->workspace,
                               a->workspace_bytes, s);
        if (rc) return rc;
    }
    if (hits) {
        rc = row_reduce<uint32_t>(hitbuf, 3, a->totalThreads, 1.0, hits, 0, a->workspace, a->workspace_bytes, s);
        if (rc) return rc;
        // hits[3] = number of event samples
        hipLaunchKernelGGL(set_double_kernel, dim3(1), dim3(1), 0, s, hits + 3, (double)a->totalThreads * (double)samples);
        MCMC_HIP_CHECK(hipGetLastError());
    }
    return MCMC_OK;
}

extern "C" int mcmc_launch_integrate_inel(const mcmc_xs_args *a, uint32_t samples, float impactParameter, double *hits2,
                                          float dmax2, void *stream)
{
    int rc = check_xs(a);
    if (rc) return rc;
    if (!hits2) return set_error(MCMC_EINVAL, "hits2 is null");
    if (samples == 0) return set_error(MCMC_EINVAL, "samples must be > 0");
    if (!(dmax2 > 0.0f)) return set_error(MCMC_EINVAL, "dmax2 must be > 0");
    if (a->nNucleonsA == 0 || a->nNucleonsB == 0) return set_error(MCMC_EINVAL, "both nuclei need at least one nucleon");
    const uint32_t nN = a->nNucleonsA + a->nNucleonsB;
    if (a->injected && a->injected_stride < (uint64_t)samples * nN * 4)
        return set_error(MCMC_EINVAL, "injected_stride too small for samples * nNucleons * 4 draws");
    if (!(a->beta > 0.0f)) return set_error(MCMC_EINVAL, "beta must be > 0");
    if (!a->workspace || a->workspace_bytes < mcmc_xs_workspace_bytes(a))
        return set_error(MCMC_ENOMEM, "workspace smaller than mcmc_xs_workspace_bytes()");
    hipStream_t s = (hipStream_t)stream;

    XsParams p = make_params(a, samples);
    if (a->math_mode == MCMC_MATH_FAST) p.pos_scale = sqrtf(1.4426950408889634f / (2.0f * a->beta));
    p.b_stored = impactParameter * p.pos_scale;
    p.hit_d2 = dmax2 * p.pos_scale * p.pos_scale;
    uint32_t *hitbuf = (uint32_t *)((char *)a->workspace + rowmean_bytes(nN < 4 ? 4 : nN));
    p.hits = hitbuf;
    const SabOptions opt = sab_options();
    const SabPlan plan = sab_plan(a->nNucleonsA, a->nNucleonsB, false, opt);
    if (plan.lds > 160 * 1024) return set_error(MCMC_EINVAL, "nuclei too large for the LDS-resident S_AB kernel (A+B <= ~1600)");
    p.ga_log2 = (uint32_t)plan.ga_log2;
    p.cull_d = sab_cull_distance(a, p, false, true, opt);
    rc = launch_sab(4, a, p, plan, a->uniform_kind == MCMC_TARGET_R2WS, s);
    if (rc) return rc;
    rc = row_reduce<uint32_t>(hitbuf, 1, a->totalThreads, 1.0, hits2, 0, a->workspace, a->workspace_bytes, s);
    if (rc) return rc;
    hipLaunchKernelGGL(set_double_kernel, dim3(1), dim3(1), 0, s, hits2 + 1, (double)a->totalThreads * (double)samples);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

extern "C" int mcmc_launch_row_mean_f32(const float *x, uint32_t rows, uint64_t cols, double *out, void *workspace,
                                        size_t workspace_bytes, void *stream)
{
    if (!x || !out || rows == 0 || cols == 0) return set_error(MCMC_EINVAL, "bad row_mean arguments");
    return row_reduce<float>(x, rows, cols, 1.0 / (double)cols, out, 0, workspace, workspace_bytes, (hipStream_t)stream);
}

// Sharding-invariant row sums for the multi-GPU reduction (row_fixed_partial_kernel, xs_kernel.hpp): no reference
// counterpart beyond the host Average it feeds (main.cu:221-223,359).
extern "C" size_t mcmc_row_sum_fixed_workspace_bytes(uint32_t rows) { return align_up((size_t)rows * RR_CHUNKS * 4 * sizeof(int64_t), 256); }

extern "C" int mcmc_launch_row_sum_fixed_f32(const float *x, uint32_t rows, uint64_t cols, double *out4, void *workspace,
                                             size_t workspace_bytes, void *stream)
{
    if (!x || !out4 || rows == 0 || cols == 0) return set_error(MCMC_EINVAL, "bad row_sum_fixed arguments");
    if (cols > (1ull << 32)) return set_error(MCMC_EINVAL, "row_sum_fixed: at most 2^32 columns (the limb sums stay exact doubles)");
    if (!workspace || workspace_bytes < mcmc_row_sum_fixed_workspace_bytes(rows))
        return set_error(MCMC_ENOMEM, "workspace smaller than mcmc_row_sum_fixed_workspace_bytes()");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(row_fixed_partial_kernel, dim3(RR_CHUNKS, rows), dim3(256), 0, s, x, cols, (int64_t *)workspace);
    MCMC_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(row_fixed_final_kernel, dim3((rows * 4 + 63) / 64), dim3(64), 0, s, (const int64_t *)workspace, out4, rows);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

// value of a limb triple {sum l2, sum l1, sum l0} (possibly summed over launches / GPUs): one rounding, from the exact integer
extern "C" double mcmc_fixed_sum_value(const double *limbs3)
{
    if (!limbs3) return 0.0;
    const __int128 q = ((__int128)(long long)limbs3[0] << (2 * FIXED_LIMB)) + ((__int128)(long long)limbs3[1] << FIXED_LIMB) +
                       (__int128)(long long)limbs3[2];
    return (double)q * 0x1p-40; // __int128 -> double rounds once; the scaling is exact
}
