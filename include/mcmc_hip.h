/*
 * mcmc_hip.h -- C ABI of libmcmc_hip.so, the MI355X (gfx950) Metropolis-Hastings
 * Monte-Carlo integration engine.
 *
 * The reference (ArktisDev/MCMC) has no plugin/FFI interface: its boundary is
 * the set of kernel launches and host statistics calls made by its two
 * drivers, stepsize/src/main.cpp and xsection/src/main.cu.
 * Every entry point below names the reference call site it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 on success, a negative MCMC_E* code for
 *     argument errors, or a positive hipError_t value; mcmc_last_error()
 *     returns a thread-local message.  The library never calls exit().
 *   - "Layer A" (mcmc_launch_*, mcmc_chain_run, ...) takes DEVICE pointers the
 *     caller owns plus a hipStream_t (as void*, NULL = default stream) and is
 *     asynchronous.  It allocates nothing.
 *   - "Layer B" (mcmc_create ... mcmc_destroy) owns its device memory and one
 *     stream, mirrors the drivers' parameter names, and is synchronous unless
 *     stated.  One handle is not thread-safe; distinct handles are.
 *   - all arrays are structure-of-arrays exactly as the reference lays them
 *     out: prevSample[nNucleons][totalThreads] (Metropolis.cuh:33,63).
 */
#ifndef MCMC_HIP_H
#define MCMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCMC_OK 0
#define MCMC_EINVAL (-1)   /* bad argument (message in mcmc_last_error) */
#define MCMC_ENOMEM (-2)   /* workspace too small / allocation failed   */
#define MCMC_ENODEV (-3)   /* no HIP device                              */

/* Target densities: the callables of xsection/include/Distributions.cuh:7-28
 * and stepsize/include/metropolis.hpp:9-17, selected at run time instead of
 * as template arguments (Common.cuh:17 `typedef float (*PDF)(float)`).     */
typedef enum {
    MCMC_TARGET_WS = 0,      /* WoodsSaxon(r, c=p0, a=p1)     */
    MCMC_TARGET_R2WS = 1,    /* R2WoodsSaxon(r, c=p0, a=p1)   */
    MCMC_TARGET_GAUSS = 2,   /* Gaussian(r, variance=p0)      */
    MCMC_TARGET_R2GAUSS = 3  /* R2Gaussian(r, variance=p0)    */
} mcmc_target_kind;

typedef struct {
    int32_t kind; /* mcmc_target_kind */
    float p0;
    float p1;
} mcmc_target;

/* Accept/reject rule. */
typedef enum {
    MCMC_ACCEPT_RECIP = 0,        /* p' * (1/p) >= U, U in (0,1]; Metropolis.cuh:47-50 (GPU rule)       */
    MCMC_ACCEPT_SHORTCIRCUIT = 1, /* p' > p || U < p'/p, U in [0,1); metropolis.hpp:41 (CPU rule, FP64) */
    MCMC_ACCEPT_NODIV = 2,        /* p' >= U * p  (same rule, no reciprocal; not bit-identical at ties) */
    MCMC_ACCEPT_LOG = 3,          /* log U <= log p' - log p (log-space)                                 */
    MCMC_ACCEPT_CROSS = 4         /* p' >= U p with p kept as a quotient n/d and the test cross-multiplied
                                     (n' d >= U n d'): the same decision with no reciprocal; FAST math only */
} mcmc_accept_rule;

typedef enum {
    MCMC_MATH_PRECISE = 0, /* OCML expf/logf/..., IEEE division, no FMA contraction */
    MCMC_MATH_FAST = 1     /* v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sin_f32 hardware ops */
} mcmc_math_mode;

#define MCMC_MAX_LAGS 49   /* stepsize/src/main.cpp:27 uses k = 1..49 */

/* Length (in doubles) of the sufficient-statistics vector of one coordinate:
 *   [0] sum y            y = x - center
 *   [1] sum y^2
 *   [2] accepted proposals
 *   [3 .. 3+L)        P_k = sum_{i<n-k} y_i y_{i+k},           k = 1..L
 *   [3+L .. 3+2L)     A_i = sum over chains of y_i,            i = 0..L-1 (first L samples of the window)
 *   [3+2L .. 3+3L)    Z_j = sum over chains of y_{n-1-j},      j = 0..L-1 (last L samples so far)
 * From these the host evaluates statistics.hpp:6-57 exactly (mcmc_stats_finalize). */
#define MCMC_STATS_LEN(n_lags) (3 + 3 * (n_lags))

/* ------------------------------------------------------------------ */
/* Layer A.1 -- many-chain MH kernel with streaming statistics.
 * Replaces MetropolisHastingsAlgorithm (stepsize/include/metropolis.hpp:28-50,
 * called from stepsize/src/main.cpp:17) and the per-lag AutoCorrelation passes
 * (statistics.hpp:36-57, main.cpp:27-30) for n_chains independent chains.   */
typedef struct {
    uint64_t n_chains;     /* chains on this device                                        */
    uint32_t n_coords;     /* independent coordinates per chain (each its own 1-D target)  */
    uint32_t n_lags;       /* 0..MCMC_MAX_LAGS lag products tracked per coordinate         */
    uint64_t n_steps;      /* sweeps in this launch (`iterations`, metropolis.hpp:28)      */
    double step_size;      /* `stepSize`, metropolis.hpp:28                                */
    double center;         /* y = x - center is accumulated (shift-invariant statistics)   */
    mcmc_target target;
    int32_t precision;     /* 32 (Metropolis.cuh arithmetic) or 64 (metropolis.hpp arithmetic) */
    int32_t accept_rule;   /* mcmc_accept_rule; 64-bit supports SHORTCIRCUIT only           */
    int32_t math_mode;     /* mcmc_math_mode (32-bit only)                                 */
    int32_t window_start;  /* 1: this launch starts a statistics window (ring = empty)      */
    uint64_t seed;         /* Philox key                                                   */
    uint64_t step_offset;  /* sweeps already done by these chains (Philox counter)         */
    uint64_t chain_offset; /* global id of this device's chain 0 (multi-GPU partition)     */
    void *state;           /* dev [n_coords][n_chains] float|double, in/out                */
    double *stats;         /* dev [n_coords][MCMC_STATS_LEN(n_lags)], overwritten           */
    void *ring;            /* dev [n_coords][n_lags][n_chains] lag history carried between launches, or NULL */
    void *samples;         /* dev [n_steps][n_coords][n_chains] sample write-back, or NULL  */
    const void *inj_normals;  /* dev [n_steps][n_coords][n_chains] injected N(0,1) draws, or NULL (Philox) */
    const void *inj_uniforms; /* dev [n_steps][n_coords][n_chains] injected uniforms (rule 1 consumes them
                                 through a per-chain cursor, like metropolis.hpp:41)         */
    uint8_t *accept_bits;  /* dev [n_steps][n_coords][n_chains] accept flags, or NULL       */
    void *workspace;       /* dev scratch for block partials                               */
    size_t workspace_bytes;
} mcmc_chain_args;

/* Bytes of workspace mcmc_chain_run needs for these arguments. */
size_t mcmc_chain_workspace_bytes(const mcmc_chain_args *a);

/* Launch (asynchronous on `stream`).
 * Lag products: the FP64 instantiation folds its partial lag products into FP64 every <= 128 steps; the FP32
 * instantiations keep P_k in FP32 for the whole launch (per chain; the sum over chains is FP64), so one launch of n
 * steps carries a relative rounding error of about sqrt(n) 2^-24 on P_k (5e-6 at n = 10^4).  Windows longer than
 * ~10^5 steps should be split into several launches (window_start = 0 + `ring`), whose statistics the host adds.
 * chain_offset + n_chains must be <= 2^40 (global chain ids are 40-bit Philox counters). */
int mcmc_chain_run(const mcmc_chain_args *a, void *stream);

/* mcmc_chain_run that allocates nothing and needs no workspace -- NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid
 * allocation).  The block partials go to a device array inside the library (MCMC_CHAIN_NOALLOC_MAX_BYTES, held by
 * every GPU the library runs on until the process ends); a->workspace / workspace_bytes are ignored.  Arguments
 * whose mcmc_chain_workspace_bytes() exceeds that size are rejected with MCMC_ENOMEM.  Results are identical to
 * mcmc_chain_run.  The array's device address is looked up (hipGetSymbolAddress) once per device, by the first
 * mcmc_chain_run OR mcmc_chain_run_noalloc call on that device: make one such call on every device you use BEFORE
 * the capture / no-allocation region, and calls inside it do nothing but launch.  All calls on one GPU share the array: calls that overlap on one GPU (different streams or
 * threads) race; calls on one stream are ordered and safe.                                                    */
#define MCMC_CHAIN_NOALLOC_MAX_BYTES (8u << 20)
int mcmc_chain_run_noalloc(const mcmc_chain_args *a, void *stream);

/* Diagnostic, no counterpart in the reference: launches the stamped twin of the kernel `a` selects (FP32, FAST math,
 * rule CROSS or RECIP, Philox draws, no write-back; the same body plus one s_memtime / s_memrealtime pair per block)
 * and returns the shader clock it sustained in GHz (median over blocks).  Advances the chains like mcmc_chain_run.
 * Synchronous; allocates its own stamp buffer.  bench.py uses it to turn kernel time into SIMD cycles.          */
int mcmc_chain_clock_probe(const mcmc_chain_args *a, double *clock_ghz, void *stream);

/* Host finalisation of a (summed) statistics vector of ONE coordinate:
 * mean, variance (n-1; statistics.hpp:17-28), acceptance ratio
 * (metropolis.hpp:49), rho_k for k=1..n_lags (statistics.hpp:36-45, pooled
 * over chains; identical to the reference for n_chains == 1), and the
 * integrated autocorrelation time 1 + 2 sum rho_k
 * (stepsize/graphing/StepSizeAnalysis.ipynb cell 1).
 * `stats` may be the sum over launches/devices of the [0..3+L) entries with
 * A from the window's first launch and Z from its last.                    */
typedef struct {
    double mean;
    double variance;
    double acceptance;
    double tau_int;
} mcmc_chain_summary;

int mcmc_stats_finalize(const double *stats, uint32_t n_lags, uint64_t n_chains, uint64_t n_steps_total,
                        double center, mcmc_chain_summary *out, double *rho /* [n_lags] or NULL */);

/* The library's random draws as device arrays [n_steps][n_chains] (any pointer
 * may be NULL), for chains chain_offset.. and coordinate / nucleon `coord`:
 *   normals, uniforms          N(0,1) proposals and accept uniforms in (0,1) of the chain kernels
 *                              (mcmc_chain_run, warm-up, R2): packed stream, 3 Philox blocks per 8 steps
 *   angle_u0, angle_u1         the two angle uniforms of the S_AB kernel
 *   sab_normals, sab_uniforms  proposals and accept uniforms in (0,1] of the S_AB kernel
 * Replaces RandomNormal/RandomUniform (stepsize/include/random.hpp:7-19) and
 * curand_normal/curand_uniform (Metropolis.cuh:43,47; IntegrateSAB.cuh:51-52):
 * Philox4x32-10 keyed by seed, counter = (chain, coord, step block, stream tag), so the
 * values do not depend on launch geometry or GPU count.                     */
int mcmc_launch_draws(uint64_t seed, uint64_t chain_offset, uint64_t n_chains, uint32_t coord, uint64_t step_offset,
                      uint64_t n_steps, int math_mode, float *normals, float *uniforms, float *angle_u0,
                      float *angle_u1, float *sab_normals, float *sab_uniforms, void *stream);

/* ------------------------------------------------------------------ */
/* Layer A.3 -- correlated Gaussian "stress" target (BASELINE.json configs[4]): n_chains chains in
 * `dim` dimensions, precision matrix symmetric tridiagonal (e.g. Sigma_ij = rho^|i-j|).  One sweep =
 * GibbsStep (xsection/include/Metropolis.cuh:69-79): dim single-coordinate MetropolisHastingsSteps
 * (:40-52) on the coordinate's conditional density, accepted in log space.  A scalar observable
 * (one coordinate) is tracked per sweep: mean / variance (statistics.hpp:6-34), and the lag-`lag`
 * autocorrelation (statistics.hpp:36-57) through a ring of the last `lag` observables per chain in
 * HBM; optional write-back of the observable or of the whole state (coalesced SoA).
 * No counterpart target exists in the reference; parity is against the oracle's restatement of the
 * same sweep under injected draws.                                                            */
#define MCMC_MVN_STATS_LEN 6 /* sum y, sum y^2, accepts, P_L = sum y_t y_{t-L}, HEAD, TAIL (first / last L samples) */
typedef struct {
    uint64_t n_chains;
    uint32_t dim;            /* multiple of 8, <= 64                                              */
    uint32_t observable;     /* coordinate whose series is tracked                                */
    uint64_t n_steps;        /* sweeps in this launch                                             */
    float step_size;         /* proposal width (Metropolis.cuh:43)                                */
    float center;            /* y = x_obs - center is accumulated                                 */
    const float *q_diag;     /* HOST [dim]   Q[i][i]                                              */
    const float *q_off;      /* HOST [dim-1] Q[i][i+1]                                            */
    int32_t math_mode;       /* mcmc_math_mode                                                    */
    uint32_t lag;            /* L >= 1                                                            */
    uint64_t seed, step_offset, chain_offset;
    uint64_t window_begin;   /* first sweep of the statistics window this launch belongs to       */
    uint64_t window_end;     /* one past its last sweep                                           */
    float *state;            /* dev [dim][n_chains], in/out                                       */
    float *ring;             /* dev [lag][n_chains]: row (t mod lag) holds y_t                    */
    float *samples_obs;      /* dev [n_steps][n_chains] observable write-back, or NULL            */
    float *samples_full;     /* dev [n_steps][dim][n_chains] state write-back, or NULL            */
    const float *inj;        /* dev [n_steps][dim][2][n_chains] injected (normal, uniform), or NULL */
    uint8_t *accept_bits;    /* dev [n_steps][dim][n_chains], or NULL                             */
    double *stats;           /* dev [MCMC_MVN_STATS_LEN], overwritten                             */
    void *workspace;
    size_t workspace_bytes;
} mcmc_mvn_args;

size_t mcmc_mvn_workspace_bytes(const mcmc_mvn_args *a);
int mcmc_mvn_run(const mcmc_mvn_args *a, void *stream);
/* Host finalisation of the window's summed statistics (statistics.hpp:6-57 for the one lag). */
int mcmc_mvn_finalize(const double *stats, uint64_t n_chains, uint32_t dim, uint64_t n_sweeps, uint32_t lag,
                      double center, mcmc_chain_summary *out, double *rho_lag);

/* ------------------------------------------------------------------ */
/* Layer A.2 -- xsection kernels (device pointers, reference layouts).   */
typedef struct {
    uint32_t nNucleonsA;        /* main.cu:43 */
    uint32_t nNucleonsB;        /* main.cu:44 */
    uint64_t totalThreads;      /* events on this device; main.cu:89 (`totalBlocks * threadsPerBlock`) */
    const mcmc_target *targets; /* dev [nNucleonsA+nNucleonsB] per-nucleon density (main.cu:47-51)    */
    float *prevSample;          /* dev [nNucleons][totalThreads]; main.cu:126                         */
    float stepsize;             /* main.cu:149,197,335 (2.2f)                                         */
    int32_t accept_rule;        /* MCMC_ACCEPT_RECIP, _NODIV, _LOG (FAST math) or _CROSS.  CROSS keeps the density as a
                                   quotient (two cached values per chain): the warm-up and R2 kernels run it as such;
                                   the S_AB / inelastic kernels hold ONE cached value per nucleon in LDS and evaluate
                                   CROSS as NODIV -- the same decision p' >= U p, identical away from exact ties (for
                                   r^2 Woods-Saxon in FAST math NODIV itself is evaluated cross-multiplied there, with
                                   e^((r-c)/a) cached and r^2 recomputed: no reciprocal) */
    int32_t math_mode;
    int32_t uniform_kind;       /* -1: kinds differ per nucleon (read from `targets` on the device);
                                   otherwise every nucleon has this mcmc_target_kind, which selects a
                                   kernel specialised for it (only MCMC_TARGET_R2WS is specialised)   */
    uint64_t seed;              /* main.cu:69 */
    uint64_t step_offset;       /* sweeps already done (replaces the curandState array, main.cu:123)  */
    uint64_t event_offset;      /* global id of this device's event 0                                 */
    const float *injected;      /* dev [totalThreads][draws_per_thread] injected draw streams in the
                                   reference's consumption order, or NULL (Philox)                    */
    uint64_t injected_stride;   /* floats per thread in `injected`                                    */
    float sigmaNN;              /* fm^2; IntegrateSAB.cuh:11 (4.3f)                                   */
    float beta;                 /* fm^2; IntegrateSAB.cuh:10 (0.2f)                                   */
    void *workspace;            /* dev scratch for reductions                                         */
    size_t workspace_bytes;
} mcmc_xs_args;

size_t mcmc_xs_workspace_bytes(const mcmc_xs_args *a);

/* InitSampleArray, Metropolis.cuh:27-35 (launched at main.cu:132).  The
 * reference's `int initSample` parameter truncates 1.0f to 1; pass the value
 * you want stored.                                                        */
int mcmc_launch_init_sample_array(float *prevSample, float initSample, uint32_t nNucleons, uint64_t totalThreads,
                                  void *stream);

/* Randomised chain starts -- what main.cu:144-148 describes but does not do
 * ("randomize the initial states of the chain ... treating it as a discrete
 * pdf, and sampling according to that"): prevSample[n][t] is drawn from nucleon
 * n's density tabulated on 1024 bins (inverse CDF, one Philox uniform of stream
 * tag 4 per chain; uses targets, prevSample, totalThreads, seed, event_offset
 * of `a`).  A short warm-up afterwards removes the tabulation error.          */
int mcmc_launch_init_sample_from_pdf(const mcmc_xs_args *a, void *stream);

/* WarmupMetropolis, Metropolis.cuh:83-108 (main.cu:149). */
int mcmc_launch_warmup_metropolis(const mcmc_xs_args *a, uint32_t iterations, void *stream);

/* MCIntegrate_R2, IntegrateR2.cuh:7-61 (main.cu:197,206).
 * resultBuffer: dev [nNucleons][totalThreads] per-thread means (as the
 * reference).  runMean: dev [nNucleons] doubles = mean over threads of each
 * row (replaces the host Average at main.cu:221-223), or NULL.             */
int mcmc_launch_integrate_r2(const mcmc_xs_args *a, float *resultBuffer, uint32_t samples, double *runMean,
                             void *stream);

/* MCIntegrate_S_AB, IntegrateSAB.cuh:19-97 (main.cu:335,344).
 * resultBuffer: dev [totalThreads]; runMean: dev [1] double (main.cu:359) or NULL.
 * hits: dev [4] doubles or NULL -- black-disk counters of the same events
 * (visualization/GlauberVisualization.ipynb cell 2, d < 2*nucleonRadius):
 *   [0] colliding pairs, [1] samples with >=1 collision, [2] participants,
 *   [3] samples (totalThreads*samples).  dmax2 = (2*nucleonRadius)^2.       */
int mcmc_launch_integrate_sab(const mcmc_xs_args *a, float *resultBuffer, uint32_t samples, float impactParameter,
                              double *runMean, double *hits, float dmax2, void *stream);

/* Black-disk inelastic flag alone, for sigma_inel scans (same events, draws and
 * chain advance as mcmc_launch_integrate_sab; the S_AB product of
 * IntegrateSAB.cuh:65-78 is not evaluated): per sample "does any A-B pair have
 * d^2 < dmax2" (visualization/GlauberVisualization.ipynb cell 2).  The pair
 * scan of an event stops at the first collision found.
 * hits2: dev [2] doubles: [0] samples with >=1 collision (identical to hits[1]
 * of mcmc_launch_integrate_sab), [1] samples (totalThreads*samples).        */
int mcmc_launch_integrate_inel(const mcmc_xs_args *a, uint32_t samples, float impactParameter, double *hits2,
                               float dmax2, void *stream);

/* Mean over each row of a [rows][cols] float matrix into out[rows] doubles
 * (device Average, Statistics.cuh:23-25; fixed-order FP64 tree).          */
int mcmc_launch_row_mean_f32(const float *x, uint32_t rows, uint64_t cols, double *out, void *workspace,
                             size_t workspace_bytes, void *stream);
size_t mcmc_row_mean_workspace_bytes(uint32_t rows, uint64_t cols);

/* Sharding-invariant sum of each row of a [rows][cols] float matrix -- the device side of the multi-GPU form of the
 * host Average of main.cu:221-223,359 (no other counterpart in the reference).  Every element (|x| < 2^14: an r^2 or
 * S_AB mean) is rounded once to a multiple of 2^-40 and the integers are summed, so the total does not depend on how
 * the columns are split over blocks, launches or GPUs.  out4: dev [rows][4] doubles = {sum l2, sum l1, sum l0, count
 * of elements that were not finite or >= 2^14}, the integer as three 20-bit limbs (exact as doubles for up to 2^32
 * columns in the whole job, so they survive a FP64 SUM all-reduce).  mcmc_fixed_sum_value (host) turns a limb triple,
 * possibly summed over launches / GPUs, into the sum with ONE rounding.                                        */
size_t mcmc_row_sum_fixed_workspace_bytes(uint32_t rows);
int mcmc_launch_row_sum_fixed_f32(const float *x, uint32_t rows, uint64_t cols, double *out4, void *workspace,
                                  size_t workspace_bytes, void *stream);
double mcmc_fixed_sum_value(const double *limbs3);

/* ------------------------------------------------------------------ */
/* Host statistics with the reference's exact arithmetic.               */
/* Statistics.cuh:7-42,63-87 (FP32 Kahan) */
float mcmc_Sum(const float *samples, int n);
float mcmc_Average(const float *samples, int n);
float mcmc_Variance(const float *samples, float avg, int n);
float mcmc_AutoCorrelation(const float *samples, float avg, int lag, int n);
/* Statistics.cuh:45-60 and :90-115 -- the overloads that compute the average themselves */
float mcmc_Variance_n(const float *samples, int n);
float mcmc_AutoCorrelation_n(const float *samples, int lag, int n);
/* statistics.hpp:6-57 (FP64 naive) */
double mcmc_Average_f64(const double *nums, int n);
double mcmc_Variance_f64(const double *nums, int n);
double mcmc_AutoCorrelation_f64(const double *nums, int n, int lag);
/* xsection/tools/XSectionAnalysis.py:3-28 -> out[6] = tot, unc, el, unc, rxn, unc (mb) */
int mcmc_cross_sections(const double *b, const double *S_AB, const double *S_AB_stderr, int n, double *out);

/* ------------------------------------------------------------------ */
/* Layer B -- handle API with the drivers' parameter names (main.cu:38-82,
 * same names as emitted by xsection/tools/MCMCMacro.py:305-368).          */
typedef struct {
    int32_t nNeutronA, nProtonA, nNeutronB, nProtonB; /* main.cu:38-41 */
    mcmc_target neutronA, protonA, neutronB, protonB; /* per-species densities (MCMCMacro.py:159-199) */
    int32_t totalBlocks;       /* main.cu:54 */
    int32_t threadsPerBlock;   /* main.cu:55 */
    int32_t samplesPerThread;  /* main.cu:56 */
    int32_t totalRuns;         /* main.cu:58 */
    int32_t samplesForMix;     /* main.cu:60 */
    float b0, db;              /* main.cu:63-64 */
    int32_t ndbs;              /* main.cu:66 */
    uint64_t seed;             /* main.cu:69 */
    float stepsize;            /* main.cu:149 (2.2f) */
    float initSample;          /* main.cu:132 (1.0f) */
    float sigmaNN, beta;       /* IntegrateSAB.cuh:10-11 */
    int32_t accept_rule, math_mode;
    int32_t device;            /* HIP device ordinal */
    uint64_t event_offset;     /* global id of this device's event 0 (multi-GPU partition) */
    int32_t integrateRMS;      /* main.cu:81 (true): mcmc_run_all runs the RMS integration             */
    int32_t integrateSAB;      /* main.cu:82 (true): mcmc_run_all runs the S_AB scan                   */
    const char *outDir;        /* main.cu:73 ("../data/"); NULL = "./"                                  */
    const char *outFile;       /* main.cu:74,77-79 ("out.root"; argv[1]); NULL = "out"; the library appends
                                  ".RMS.csv" / ".S_AB.csv" (TTree columns of IntegrationOutputHandler.cuh:83-112) */
    const mcmc_target *nucleon_targets; /* NULL, or one density per NUCLEON in pdfArr order -- protons A, neutrons A,
                                  protons B, neutrons B (MCMCMacro.py:315-345; SetNeutronPDFA(pdf, i) etc., :172-199) --
                                  nNucleonsA + nNucleonsB entries (host memory, read by mcmc_create /
                                  mcmc_multi_create only); overrides the four per-species densities above */
} mcmc_config;

typedef struct mcmc_handle_s *mcmc_handle;

void mcmc_config_default(mcmc_config *cfg);   /* the constants of main.cu:38-82; start every config from it (or from zeros):
                                                 optional fields at the end of the struct mean "not used" only when zero */
int mcmc_create(const mcmc_config *cfg, mcmc_handle *out);   /* cudaMalloc + RandStateInit + InitSampleArray, main.cu:123-134 */
int mcmc_destroy(mcmc_handle h);                              /* main.cu:419-422 */
int mcmc_warmup(mcmc_handle h, float stepsize, int iterations);   /* main.cu:149-152 */
int mcmc_init_from_pdf(mcmc_handle h);   /* randomised starts (main.cu:144-148), see mcmc_launch_init_sample_from_pdf */
/* One run of MCIntegrate_R2 over all threads: out_run_mean[nNucleons] (main.cu:197-223). */
int mcmc_integrate_r2(mcmc_handle h, float stepsize, int samples, float *out_run_mean);
/* One run of MCIntegrate_S_AB at impact parameter b (main.cu:335-359); hits may be NULL. */
int mcmc_integrate_sab(mcmc_handle h, float stepsize, int samples, float b, float *out_run_mean, double *hits4,
                       float dmax2);
/* Whole RMS integration (main.cu:157-286): totalRuns runs, then per nucleon
 * RMS and RMS_stderr (main.cu:253-258).  out arrays [nNucleons].            */
int mcmc_run_rms(mcmc_handle h, float *rms, float *rms_stderr);
/* Whole S_AB scan (main.cu:289-413): ndbs impact parameters; out arrays [ndbs]:
 * b, S_AB, S_AB_stderr (main.cu:384-389); p_inel may be NULL (black-disk
 * inelastic probability per b when dmax2 > 0).                              */
int mcmc_run_sab_scan(mcmc_handle h, float *b, float *S_AB, float *S_AB_stderr, double *p_inel, float dmax2);
/* The same scan (main.cu:289-413: ndbs impact parameters x totalRuns runs) for the
 * black-disk inelastic probability alone: p_inel[ndbs] and its run-to-run standard
 * error (the estimator of main.cu:384-386 applied to the per-run probabilities; 0
 * when totalRuns == 1).                                                          */
int mcmc_run_inel_scan(mcmc_handle h, float *b, double *p_inel, double *p_inel_stderr, float dmax2);
/* The body of main() after the warm-up (main.cu:157-416) as the config describes it: the RMS integration when
 * cfg.integrateRMS, the S_AB scan when cfg.integrateSAB, each written to <outDir><outFile>.RMS.csv / .S_AB.csv with
 * the reference's TTree column names.  sigma6 (may be NULL) receives mcmc_cross_sections of the scan.           */
int mcmc_run_all(mcmc_handle h, double *sigma6);
/* Copy the chain state (prevSample, [nNucleons][totalThreads]) to the host. */
int mcmc_get_state(mcmc_handle h, float *host_out);
int mcmc_wait(mcmc_handle h);

/* stepsize driver in one call: n_chains chains of n_steps sweeps at `step`,
 * after n_warmup discarded sweeps; acceptance, rho_k (k=1..n_lags) and
 * optional samples [n_steps][n_chains] (host).  precision/rule as
 * mcmc_chain_args.  Replaces main.cpp:17-30 for many chains.               */
int mcmc_stepsize_run(const mcmc_target *target, int precision, int accept_rule, int math_mode, double step,
                      uint64_t n_chains, uint32_t n_coords, uint64_t n_warmup, uint64_t n_steps, uint32_t n_lags,
                      uint64_t seed, double x0, mcmc_chain_summary *summary /* [n_coords] */,
                      double *rho /* [n_coords][n_lags] or NULL */, void *samples_host /* or NULL */);

/* The step-size study (stepsize/src/main.cpp:14-36; efficiency 1/(1 + 2 sum rho_k) of
 * StepSizeAnalysis.ipynb cell 1) on the device: n_widths equally spaced widths in
 * [lo, hi], n_chains chains of n_warmup + n_steps steps each (FP32, 49 lags); out
 * arrays [n_widths] (any may be NULL); *best_width = width of least integrated
 * autocorrelation time (parabola through the grid minimum).  Use it for the width
 * that main.cu:149,197,335 hard-codes as 2.2f.                                   */
int mcmc_tune_stepsize(const mcmc_target *target, double lo, double hi, int n_widths, uint64_t n_chains,
                       uint64_t n_warmup, uint64_t n_steps, uint64_t seed, double x0, double *widths,
                       double *acceptance, double *tau_int, double *best_width);

/* ------------------------------------------------------------------ */
/* Layer C -- the two drivers on several GPUs of ONE node (north_star: "chains are partitioned across the 8 GPUs ...
 * only sufficient statistics are combined with one RCCL all-reduce per batch").  The reference is single-GPU; its
 * analogue of the exchange is the host Average over threads and runs (main.cu:221-223,253-258,359,384-389).
 * A "shard" is a contiguous, balanced range of GLOBAL event / chain ids on one device (shard r of W owns
 * [r*base + min(r, extra), ...) with base = n / W, extra = n % W: the first `extra` shards hold one id more); every
 * shard queues all its launches on its own stream from its own host thread and leaves its sufficient statistics in
 * one device vector; ONE all-reduce (SUM, FP64) of that vector ends the batch.
 *   MCMC_REDUCE_RCCL  librccl, loaded at run time: ncclCommInitAll over the shards' devices + one grouped
 *                     ncclAllReduce(double, sum) on the shards' streams (needs one device per shard);
 *   MCMC_REDUCE_HOST  every shard copies its vector back and the host adds them in shard order;
 *   MCMC_REDUCE_AUTO  RCCL when librccl loads, there are >= 2 shards and no two share a device; else the host.
 * Several shards may name the same device (a one-GPU box rehearsing the partition; host reduction only).
 * xsection payloads are integers held as doubles (hit counts; S_AB / r^2 sums as fixed-point limbs,
 * mcmc_launch_row_sum_fixed_f32): every result of mcmc_multi_* is bit-identical for any number of shards.      */
#define MCMC_REDUCE_AUTO (-1)
#define MCMC_REDUCE_HOST 0
#define MCMC_REDUCE_RCCL 1

typedef struct mcmc_multi_s *mcmc_multi;

/* cfg as for mcmc_create: cfg->totalBlocks * cfg->threadsPerBlock is the WHOLE job's event count (main.cu:89),
 * partitioned over devices[0 .. n_shards); cfg->device is ignored, cfg->event_offset is the job's first id.    */
int mcmc_multi_create(const mcmc_config *cfg, const int *devices, int n_shards, int reduce, mcmc_multi *out);
int mcmc_multi_destroy(mcmc_multi m);
/* n_shards, the reduction in use (MCMC_REDUCE_HOST / _RCCL) and each shard's first id / id count (any may be NULL) */
int mcmc_multi_info(mcmc_multi m, int *n_shards, int *reduce_used, uint64_t *offsets, uint64_t *counts);
/* the Layer B handle of one shard (owned by `m`; e.g. for mcmc_get_state), or NULL */
mcmc_handle mcmc_multi_shard(mcmc_multi m, int shard);
int mcmc_multi_warmup(mcmc_multi m, float stepsize, int iterations);   /* main.cu:149-152 on every shard */
int mcmc_multi_init_from_pdf(mcmc_multi m);                            /* mcmc_init_from_pdf on every shard */
/* mcmc_run_rms / mcmc_run_sab_scan / mcmc_run_inel_scan over all shards: the same outputs, evaluated on the
 * all-reduced sums (the run mean of main.cu:221-223,359 is total / totalThreads of the whole job).            */
int mcmc_multi_run_rms(mcmc_multi m, float *rms, float *rms_stderr);
int mcmc_multi_run_sab_scan(mcmc_multi m, float *b, float *S_AB, float *S_AB_stderr, double *p_inel, float dmax2);
int mcmc_multi_run_inel_scan(mcmc_multi m, float *b, double *p_inel, double *p_inel_stderr, float dmax2);

/* mcmc_stepsize_run with the chains partitioned over devices[0 .. n_shards): one all-reduce of the
 * [launches][n_coords][MCMC_STATS_LEN(n_lags)] statistics (FP64 sums over chains: equal across partitions to
 * FP64 rounding).  samples_host needs n_shards == 1.  mcmc_stepsize_run is this call on the current device.
 * FP32 windows longer than 65536 steps run as ring-carried launches of at most that length (the FP32 kernels
 * keep their lag products in FP32 per launch, see mcmc_chain_run) whose statistics are added in FP64; the
 * accumulators are centred on the target's mean (quadrature), not on x0.                                      */
int mcmc_stepsize_run_multi(const mcmc_target *target, int precision, int accept_rule, int math_mode, double step,
                            uint64_t n_chains, uint32_t n_coords, uint64_t n_warmup, uint64_t n_steps, uint32_t n_lags,
                            uint64_t seed, double x0, const int *devices, int n_shards, int reduce,
                            mcmc_chain_summary *summary /* [n_coords] */, double *rho /* [n_coords][n_lags] or NULL */,
                            void *samples_host /* or NULL */);

/* Isotope name ("PB-208", "208-pb") -> Z and A: NuclearZAndA, xsection/tools/MCMCMacro.py:63-83
 * (used by IntegrationSettings.SetIsotopeA/B, :145-157: protons = Z, neutrons = A - Z). */
int mcmc_isotope(const char *name, int *Z, int *A);

/* Diagnostic switches of the S_AB kernels, no counterpart in the reference (INTEGRATION.md "Environment switches"):
 * "MCMC_SAB_CULL" ("0" = exact pair culling off), "MCMC_SAB_SMALL" ("0" / "1" = general / register kernel for nuclei of
 * at most 4 + 4 nucleons), "MCMC_SAB_PLAN" ("G,ga_log2").  The environment variables of the same names are read once,
 * at the first S_AB launch; this call overrides them for the process (value NULL = default).  Results do not depend
 * on them beyond the order of float multiplications.
 * Chain kernels: "MCMC_MVN_REG" ("0" / "1" = LDS / register kernel of the stress target), "MCMC_LAG_MFMA" ("0" = the
 * 49 lag products of FP32 chains on the vector unit instead of the matrix cores: same chains, lag sums equal to
 * FP32 rounding), "MCMC_SAMPLES_PAIRS" ("1" = mcmc_chain_args.samples is written as [n_steps / 2][n_coords][n_chains][2],
 * two consecutive steps of a chain side by side -- one 64-bit store per lane and step pair; FP32 FAST-math Philox
 * launches without lags or accept flags, step_offset and n_steps even, else MCMC_EINVAL; the values are those of the
 * default layout bit for bit).                                                                                   */
int mcmc_set_option(const char *name, const char *value);

const char *mcmc_last_error(void);
const char *mcmc_version(void);
/* Rounds of the Philox4x32 generator this build of the library was compiled with: 10 (default; every parity test) or
 * 7 (`make philox7` -> libmcmc_hip_philox7.so: the fewest rounds that pass BigCrush, Salmon et al. SC'11; ~10 % more
 * proposals/s).  The draw streams of the two builds differ; each is pinned by Random123's known-answer vectors.   */
int mcmc_philox_rounds(void);
int mcmc_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* MCMC_HIP_H */
