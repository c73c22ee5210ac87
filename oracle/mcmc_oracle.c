/*
 * mcmc_oracle.c -- CPU restatement of the ArktisDev/MCMC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (mcmc_amd/, include/,
 * bench.py's GPU legs) may link, import or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / CPU baseline.  Build: see oracle/Makefile (gcc -O2
 * -ffp-contract=off, so no FMA contraction alters the FP32/FP64 rounding).
 *
 * Every function cites the reference lines (relative to /root/reference) whose
 * algorithm it restates.  Parity pinning:
 *   - stepsize path (a1-a4): pinned against the reference's own headers
 *     compiled into oracle/_ref (oracle/ref_stepsize_shim.cpp) -- see
 *     tests/test_oracle_pinning.py and tests/golden/.
 *   - xsection path (a5-a15): the .cuh files need the CUDA toolkit and cannot
 *     be built here; the reference holds no golden vectors for them, so these
 *     functions are a line-by-line restatement.  Draw-level parity is unpinned
 *     against reference outputs ("parity unpinned"); at the level of estimates
 *     they are anchored on the one output the reference records (the C-12 + C-12
 *     cross sections and S_AB table in xsection/tools/XSectionAnalysis.ipynb,
 *     tests/golden/reference_c12_notebook.json, with the Woods-Saxon radius
 *     inferred) and on closed-form / quadrature known answers.
 *   - Philox4x32-10: third-party published algorithm (Salmon et al., SC'11,
 *     Random123 v1.09), pinned by Random123's known-answer vectors.
 *   - sigma analysis (f1): pinned against xsection/tools/XSectionAnalysis.py
 *     run in-container (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* Philox4x32-10 (Random123).  Not in the reference: the reference uses
 * cuRAND XORWOW (xsection/include/Metropolis.cuh:23,43,47), which is
 * third-party and unpinned; the build replaces it by this counter RNG.  */
/* Rounds: 10 by default (Philox4x32-10); orc_set_philox_rounds(7) makes every draw function of this file follow
 * the library's MCMC_PHILOX_ROUNDS=7 build (Philox4x32-7, also a Random123 configuration with its own vectors). */
static int orc_philox_rounds = 10;
ORC_API void orc_set_philox_rounds(int rounds) { orc_philox_rounds = rounds; }

ORC_API void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < orc_philox_rounds; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Library draw convention (product spec, DESIGN.md "RNG"): independent
 * restatement used to cross-check the device code.
 *   key = (seed_lo, seed_hi)
 *   ctr = (id_lo, (coord << 8) | id_hi8, blk_lo, (tag << 24) | blk_hi24)
 * tag 0: MH draws of the chain kernels, packed: 3 blocks per group of 8 steps.
 * tag 1: angle draws of the S_AB kernel, blk = step >> 1.
 * tag 2: MH draws of the S_AB kernel, blk = step >> 1.                      */
static void orc_block(uint64_t seed, uint64_t id, uint32_t coord, uint64_t blk, uint32_t tag, uint32_t out[4])
{
    uint32_t ctr[4], key[2];
    ctr[0] = (uint32_t)id;
    ctr[1] = (coord << 8) | (uint32_t)((id >> 32) & 0xFFu);
    ctr[2] = (uint32_t)blk;
    ctr[3] = (tag << 24) | (uint32_t)((blk >> 32) & 0xFFFFFFu);
    key[0] = (uint32_t)seed;
    key[1] = (uint32_t)(seed >> 32);
    orc_philox4x32_10(ctr, key, out);
}

/* (0,1] like curand_uniform (Metropolis.cuh:47 relies on U never being 0). */
static float orc_u01_f32(uint32_t r)
{
    return fmaf((float)r, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

/* Packed MH draws (tag 0; tag 3 for the stress target).  Steps are grouped by 8
 * (group g = t >> 3), a group into 4 step pairs j = (t >> 1) & 3.  Blocks
 * B_0..B_2 = Philox(blk = 3 g + j) serve pairs 0..2: each of the 4 words gives a
 * 23-bit field from its LOW 23 bits.  The top bytes of the 12 words make the four
 * fields of pair 3:
 *   field j (j = 0..2) = low 23 bits of  B_j.w0.b3 | B_j.w1.b3 << 8 | B_j.w2.b3 << 16,
 *   field 3            = low 23 bits of  B_0.w3.b3 | B_1.w3.b3 << 8 | B_2.w3.b3 << 16.
 * So 384 Philox bits carry 16 draws of 23 bits instead of 12 of 32.
 * Field m -> uniform (2m + 1) 2^-24 in (0,1) (never 0, never 1); the
 * Box-Muller angle uses the fraction m 2^-23 itself (a grid symmetric under
 * theta -> theta + 1/2, so proposals stay exactly symmetric).
 * Pair fields: m[0] radius, m[1] angle, m[2] / m[3] accept uniform of the
 * even / odd step.                                                         */
static void orc_group_fields(uint64_t seed, uint64_t id, uint32_t coord, uint64_t g, uint32_t tag, uint32_t m[4][4])
{
    uint32_t w[3][4];
    for (uint32_t j = 0; j < 3; ++j) {
        orc_block(seed, id, coord, 3 * g + j, tag, w[j]);
        for (int k = 0; k < 4; ++k) m[j][k] = w[j][k] & 0x7FFFFFu;
        m[3][j] = ((w[j][0] >> 24) | ((w[j][1] >> 24) << 8) | ((w[j][2] >> 24) << 16)) & 0x7FFFFFu;
    }
    m[3][3] = ((w[0][3] >> 24) | ((w[1][3] >> 24) << 8) | ((w[2][3] >> 24) << 16)) & 0x7FFFFFu;
}

static void orc_pair_fields_tag(uint64_t seed, uint64_t id, uint32_t coord, uint64_t t, uint32_t tag, uint32_t m[4])
{
    uint32_t all[4][4];
    orc_group_fields(seed, id, coord, t >> 3, tag, all);
    memcpy(m, all[(t >> 1) & 3u], sizeof(uint32_t) * 4);
}

static void orc_pair_fields(uint64_t seed, uint64_t id, uint32_t coord, uint64_t t, uint32_t m[4])
{
    orc_pair_fields_tag(seed, id, coord, t, 0u, m);
}

static float orc_field_uniform(uint32_t m) { return (float)(2 * m + 1) * 5.9604644775390625e-08f; } /* exact */

/* MH draws for step t of chain (id, coord): normal and accept uniform. */
ORC_API void orc_mh_draws_f32(uint64_t seed, uint64_t id, uint32_t coord, uint64_t t, float *normal, float *uniform)
{
    uint32_t m[4];
    orc_pair_fields(seed, id, coord, t, m);
    float u1 = orc_field_uniform(m[0]);
    float rad = sqrtf(-2.0f * logf(u1));
    double ang = 2.0 * M_PI * ((double)m[1] * 1.1920928955078125e-07);
    if (t & 1u) {
        *normal = rad * (float)sin(ang);
        *uniform = orc_field_uniform(m[3]);
    } else {
        *normal = rad * (float)cos(ang);
        *uniform = orc_field_uniform(m[2]);
    }
}

/* FP64 flavour for the stepsize (CPU-rule) kernels: the same fields, Box-Muller
 * in double, uniform m 2^-23 in [0,1) like std::uniform_real_distribution
 * (stepsize/include/random.hpp:10).                                       */
ORC_API void orc_mh_draws_f64(uint64_t seed, uint64_t id, uint32_t coord, uint64_t t, double *normal, double *uniform)
{
    uint32_t m[4];
    orc_pair_fields(seed, id, coord, t, m);
    double u1 = (double)orc_field_uniform(m[0]);
    double rad = sqrt(-2.0 * log(u1));
    double ang = 2.0 * M_PI * ((double)m[1] * 1.1920928955078125e-07);
    if (t & 1u) {
        *normal = rad * sin(ang);
        *uniform = (double)m[3] * 1.1920928955078125e-07;
    } else {
        *normal = rad * cos(ang);
        *uniform = (double)m[2] * 1.1920928955078125e-07;
    }
}

/* S_AB kernel streams: one block per step pair and per tag, 32-bit uniforms.
 * Box-Muller over (r0, r1): even step gets the cosine branch, odd the sine. */
ORC_API void orc_sab_mh_draws_f32(uint64_t seed, uint64_t id, uint32_t coord, uint64_t t, float *normal, float *uniform)
{
    uint32_t r[4];
    orc_block(seed, id, coord, t >> 1, 2u, r);
    float u1 = orc_u01_f32(r[0]);
    float u2 = orc_u01_f32(r[1]);
    float rad = sqrtf(-2.0f * logf(u1));
    double ang = 2.0 * M_PI * (double)u2;
    if (t & 1u) {
        *normal = rad * (float)sin(ang);
        *uniform = orc_u01_f32(r[3]);
    } else {
        *normal = rad * (float)cos(ang);
        *uniform = orc_u01_f32(r[2]);
    }
}

ORC_API void orc_angle_draws_f32(uint64_t seed, uint64_t id, uint32_t coord, uint64_t t, float *u0, float *u1)
{
    uint32_t r[4];
    orc_block(seed, id, coord, t >> 1, 1u, r);
    if (t & 1u) { *u0 = orc_u01_f32(r[2]); *u1 = orc_u01_f32(r[3]); }
    else        { *u0 = orc_u01_f32(r[0]); *u1 = orc_u01_f32(r[1]); }
}

/* ------------------------------------------------------------------ */
/* Target densities.                                                   */
enum { ORC_WS = 0, ORC_R2WS = 1, ORC_GAUSS = 2, ORC_R2GAUSS = 3 };

/* stepsize/include/metropolis.hpp:9-12 */
ORC_API double orc_R2WoodsSaxon_f64(double r, double c, double a)
{
    if (r < 0.0) return 0.0;
    return r * r / (1 + exp((r - c) / a));
}

/* stepsize/include/metropolis.hpp:14-17 */
ORC_API double orc_R2Gaussian_f64(double r, double variance)
{
    if (r < 0.0) return 0.0;
    return r * r * exp(-r * r / (2 * variance));
}

/* xsection/include/Distributions.cuh:7-10 */
ORC_API float orc_WoodsSaxon_f32(float r, float c, float a)
{
    if (r < 0.0f) return 0.0f;
    return 1 / (1 + expf((r - c) / a));
}

/* xsection/include/Distributions.cuh:13-16 */
ORC_API float orc_R2WoodsSaxon_f32(float r, float c, float a)
{
    if (r < 0.0f) return 0.0f;
    return r * r / (1 + expf((r - c) / a));
}

/* xsection/include/Distributions.cuh:19-22 */
ORC_API float orc_Gaussian_f32(float r, float variance)
{
    if (r < 0.0f) return 0.0f;
    return expf(-r * r / (2 * variance));
}

/* xsection/include/Distributions.cuh:25-28 */
ORC_API float orc_R2Gaussian_f32(float r, float variance)
{
    if (r < 0.0f) return 0.0f;
    return r * r * expf(-r * r / (2 * variance));
}

static double orc_pdf_f64(int kind, double p0, double p1, double r)
{
    switch (kind) {
    case ORC_R2WS: return orc_R2WoodsSaxon_f64(r, p0, p1);
    case ORC_R2GAUSS: return orc_R2Gaussian_f64(r, p0);
    case ORC_WS: return (r < 0.0) ? 0.0 : 1 / (1 + exp((r - p0) / p1));
    case ORC_GAUSS: return (r < 0.0) ? 0.0 : exp(-r * r / (2 * p0));
    }
    return 0.0;
}

static float orc_pdf_f32(int kind, float p0, float p1, float r)
{
    switch (kind) {
    case ORC_R2WS: return orc_R2WoodsSaxon_f32(r, p0, p1);
    case ORC_R2GAUSS: return orc_R2Gaussian_f32(r, p0);
    case ORC_WS: return orc_WoodsSaxon_f32(r, p0, p1);
    case ORC_GAUSS: return orc_Gaussian_f32(r, p0);
    }
    return 0.0f;
}

ORC_API float orc_target_f32(int kind, float p0, float p1, float r) { return orc_pdf_f32(kind, p0, p1, r); }
ORC_API double orc_target_f64(int kind, double p0, double p1, double r) { return orc_pdf_f64(kind, p0, p1, r); }

/* ------------------------------------------------------------------ */
/* a1: stepsize/include/metropolis.hpp:28-50 (one FP64 chain).
 * The density of the current point is re-evaluated every iteration (:38) and
 * a uniform is consumed only when the proposal is not uphill (:41, short-
 * circuit ||).  normals[i] replaces RandomNormal(0,1) (:36), uniforms[cursor]
 * replaces RandomUniform(0,1).  Returns the number of accepted proposals;
 * *n_uniform_used = how many uniforms were consumed.                      */
ORC_API int orc_mh_chain_f64(int iterations, double stepSize, int kind, double p0, double p1, double x0,
                             const double *normals, const double *uniforms,
                             double *samples, uint8_t *accept_bits, int *n_uniform_used)
{
    double r = x0;
    int accepted = 0;
    int cursor = 0;
    for (int i = 0; i < iterations; ++i) {
        double proposed = r + stepSize * normals[i];
        double curPdf = orc_pdf_f64(kind, p0, p1, r);
        double proposedPdf = orc_pdf_f64(kind, p0, p1, proposed);
        int take = proposedPdf > curPdf;
        if (!take) {
            double u = uniforms[cursor++];
            take = u < proposedPdf / curPdf;
        }
        if (take) { r = proposed; accepted += 1; }
        if (samples) samples[i] = r;
        if (accept_bits) accept_bits[i] = (uint8_t)take;
    }
    if (n_uniform_used) *n_uniform_used = cursor;
    return accepted;
}

/* Same chain, draws from the library's Philox convention (FP64 flavour);
 * the uniform of step i is drawn at a fixed counter whether or not it is
 * needed, so short-circuiting changes nothing observable.                  */
ORC_API int orc_mh_chain_f64_philox(int iterations, double stepSize, int kind, double p0, double p1, double x0,
                                    uint64_t seed, uint64_t chain, uint32_t coord, uint64_t step_offset,
                                    double *samples, uint8_t *accept_bits)
{
    double r = x0;
    int accepted = 0;
    for (int i = 0; i < iterations; ++i) {
        double n, u;
        orc_mh_draws_f64(seed, chain, coord, step_offset + (uint64_t)i, &n, &u);
        double proposed = r + stepSize * n;
        double curPdf = orc_pdf_f64(kind, p0, p1, r);
        double proposedPdf = orc_pdf_f64(kind, p0, p1, proposed);
        int take = proposedPdf > curPdf || u < proposedPdf / curPdf;
        if (take) { r = proposed; accepted += 1; }
        if (samples) samples[i] = r;
        if (accept_bits) accept_bits[i] = (uint8_t)take;
    }
    return accepted;
}

/* ------------------------------------------------------------------ */
/* a4: stepsize/include/statistics.hpp:6-57 (naive FP64 sums).          */
ORC_API double orc_average_f64(const double *nums, int n)
{
    double sum = 0;
    for (int i = 0; i < n; ++i) sum += nums[i];
    return sum / n;
}

ORC_API double orc_variance_f64(const double *nums, int n, double average)
{
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        double d = nums[i] - average;
        sum += d * d;
    }
    return sum / (n - 1);
}

ORC_API double orc_autocorrelation_f64(const double *nums, int n, int lag)
{
    double avg = orc_average_f64(nums, n);
    double var = orc_variance_f64(nums, n, avg);
    double sum = 0;
    for (int i = 0; i < n - lag; ++i) sum += (nums[i] - avg) * (nums[i + lag] - avg);
    return sum / (n - lag - 1) / var;
}

/* ------------------------------------------------------------------ */
/* a14: xsection/include/Statistics.cuh:7-115 (FP32 Kahan, host loops).  */
ORC_API float orc_sum_f32(const float *samples, int n)
{
    float res = 0.0f, err = 0.0f;
    for (int i = 0; i < n; ++i) {
        float y = samples[i] - err;
        float t = res + y;
        err = (t - res) - y;
        res = t;
    }
    return res - err;
}

ORC_API float orc_average_f32(const float *samples, int n) { return orc_sum_f32(samples, n) / n; }

/* Statistics.cuh:28-42 -- population variance (divides by n). */
ORC_API float orc_variance_f32(const float *samples, float avg, int n)
{
    float res = 0.0f, err = 0.0f;
    for (int i = 0; i < n; ++i) {
        float y = samples[i] - avg;
        y = y * y - err;
        float t = res + y;
        err = (t - res) - y;
        res = t;
    }
    return res / n;
}

/* Statistics.cuh:63-87 -- num over n-lag terms, den over all n terms. */
ORC_API float orc_autocorrelation_f32(const float *samples, float avg, int lag, int n)
{
    float num = 0.0f, den = 0.0f, err = 0.0f;
    for (int i = 0; i < n - lag; ++i) {
        float y = (samples[i] - avg) * (samples[i + lag] - avg) - err;
        float t = num + y;
        err = (t - num) - y;
        num = t;
    }
    err = 0.0f;
    for (int i = 0; i < n; ++i) {
        float y = (samples[i] - avg) * (samples[i] - avg) - err;
        float t = den + y;
        err = (t - den) - y;
        den = t;
    }
    return num / den;
}

/* Statistics.cuh:45-60 -- Variance(float*, int): the average first (:46), then the same Kahan loop. */
ORC_API float orc_variance_n_f32(const float *samples, int n)
{
    float avg = orc_average_f32(samples, n);
    float res = 0.0f, err = 0.0f;
    for (int i = 0; i < n; ++i) {
        float y = samples[i] - avg;
        y = y * y - err;
        float t = res + y;
        err = (t - res) - y;
        res = t;
    }
    return res / n;
}

/* Statistics.cuh:90-115 -- AutoCorrelation(float*, int lag, int n): the average first (:91). */
ORC_API float orc_autocorrelation_n_f32(const float *samples, int lag, int n)
{
    float avg = orc_average_f32(samples, n);
    float num = 0.0f, den = 0.0f, err = 0.0f;
    for (int i = 0; i < n - lag; ++i) {
        float y = (samples[i] - avg) * (samples[i + lag] - avg) - err;
        float t = num + y;
        err = (t - num) - y;
        num = t;
    }
    err = 0.0f;
    for (int i = 0; i < n; ++i) {
        float y = (samples[i] - avg) * (samples[i] - avg) - err;
        float t = den + y;
        err = (t - den) - y;
        den = t;
    }
    return num / den;
}

/* ------------------------------------------------------------------ */
/* a5-a8: one reference GPU "thread" = one event of nN radial coordinates.
 * A thread's draw stream is consumed in the reference order: per sweep, per
 * nucleon: normal then uniform (Metropolis.cuh:43,47,77); in the S_AB kernel
 * followed, per nucleon, by u0 then u1 (IntegrateSAB.cuh:50-52).          */
typedef struct {
    int kind;
    float p0;
    float p1;
} orc_target;

/* Metropolis.cuh:40-52 */
static void orc_mh_step_f32(const orc_target *tg, float *x, float *invP, float step, float normal, float uniform, uint8_t *bit)
{
    float next = *x + step * normal;
    float nextPdf = orc_pdf_f32(tg->kind, tg->p0, tg->p1, next);
    int take = nextPdf * (*invP) >= uniform;
    if (take) {
        *x = next;
        *invP = 1 / nextPdf;
    }
    if (bit) *bit = (uint8_t)take;
}

/* Metropolis.cuh:83-108 (WarmupMetropolis, one thread).
 * draws: iterations * nN * 2 floats.  accept_bits optional [iterations*nN]. */
ORC_API void orc_warmup_thread_trace(int nN, const orc_target *targets, float *x, float step, int iterations,
                                     const float *draws, uint8_t *accept_bits, float *trace, float *min_margin);

ORC_API void orc_warmup_thread(int nN, const orc_target *targets, float *x, float step, int iterations,
                               const float *draws, uint8_t *accept_bits)
{
    orc_warmup_thread_trace(nN, targets, x, step, iterations, draws, accept_bits, NULL, NULL);
}

/* Same, also recording x after every sweep (trace[iterations*nN]) and the
 * smallest |p' * invP - U| met (how far the run stayed from an accept/reject
 * tie; a device run whose expf differs in the last ulp can only disagree on
 * the accept bit when this margin is of the order of 1e-7).                */
ORC_API void orc_warmup_thread_trace(int nN, const orc_target *targets, float *x, float step, int iterations,
                                     const float *draws, uint8_t *accept_bits, float *trace, float *min_margin)
{
    float mm = 1e30f;
    float *invP = (float *)malloc(sizeof(float) * (size_t)nN);
    for (int n = 0; n < nN; ++n) invP[n] = 1.f / orc_pdf_f32(targets[n].kind, targets[n].p0, targets[n].p1, x[n]); /* :64 */
    for (int i = 0; i < iterations; ++i) {
        for (int n = 0; n < nN; ++n) {
            const float *d = draws + ((size_t)i * nN + n) * 2;
            if (min_margin) {
                float next = x[n] + step * d[0];
                float m = fabsf(orc_pdf_f32(targets[n].kind, targets[n].p0, targets[n].p1, next) * invP[n] - d[1]);
                if (m < mm) mm = m;
            }
            orc_mh_step_f32(&targets[n], &x[n], &invP[n], step, d[0], d[1],
                            accept_bits ? &accept_bits[(size_t)i * nN + n] : NULL);
            if (trace) trace[(size_t)i * nN + n] = x[n];
        }
    }
    if (min_margin) *min_margin = mm;
    free(invP);
}

/* IntegrateR2.cuh:7-61 (one thread).  out[nN] = per-nucleon mean of r^2.
 * draws: samples * nN * 2 floats.                                         */
ORC_API void orc_integrate_r2_thread(int nN, const orc_target *targets, float *x, float step, int samples,
                                     const float *draws, float *out)
{
    float *invP = (float *)malloc(sizeof(float) * (size_t)nN * 3);
    float *sum = invP + nN, *err = sum + nN;
    for (int n = 0; n < nN; ++n) {
        invP[n] = 1.f / orc_pdf_f32(targets[n].kind, targets[n].p0, targets[n].p1, x[n]);
        sum[n] = 0.0f;
        err[n] = 0.0f;
    }
    for (int s = 0; s < samples; ++s) {
        for (int n = 0; n < nN; ++n) {
            const float *d = draws + ((size_t)s * nN + n) * 2;
            orc_mh_step_f32(&targets[n], &x[n], &invP[n], step, d[0], d[1], NULL);
        }
        for (int n = 0; n < nN; ++n) { /* :38-46 */
            float y = x[n] * x[n];
            y -= err[n];
            float t = sum[n] + y;
            err[n] = (t - sum[n]) - y;
            sum[n] = t;
        }
    }
    for (int n = 0; n < nN; ++n) out[n] = (sum[n] - err[n]) / samples; /* :51 */
    free(invP);
}

static const float ORC_PI_F = 3.141592653589f; /* Common.cuh:21 */

/* IntegrateSAB.cuh:9-13 with sigmaNN, beta as parameters (reference: 4.3f, 0.2f). */
ORC_API float orc_GammaNNR2(float r, float sigmaNN, float beta)
{
    return sigmaNN / (4 * ORC_PI_F * beta) * expf(-r / (2.f * beta));
}

/* IntegrateSAB.cuh:15-17 */
ORC_API float orc_GammaNN(float dx, float dy, float sigmaNN, float beta)
{
    return orc_GammaNNR2(dx * dx + dy * dy, sigmaNN, beta);
}

static float orc_cospif(float x) { return (float)cos(M_PI * (double)x); }
static float orc_sinpif(float x) { return (float)sin(M_PI * (double)x); }

/* IntegrateSAB.cuh:19-97 (one thread).  Returns the thread's Kahan mean of
 * y = prod_{i<nA} prod_{j<nB} (1 - Gamma).  draws: samples * nN * 4 floats:
 * per sample [2*nN MH draws][2*nN angle draws].  If hits != NULL also counts,
 * with the black-disk rule of visualization/GlauberVisualization.ipynb cell 2
 * (CalculateInteractionMatrix: d < 2*nucleonRadius; here d^2 < dmax2),
 * hits[0] += number of colliding pairs, hits[1] += samples with >= 1
 * collision, hits[2] += participants (nucleons with >= 1 collision).      */
ORC_API float orc_integrate_sab_thread(int nA, int nB, const orc_target *targets, float *x, float step, int samples,
                                       float b, float sigmaNN, float beta, const float *draws,
                                       float dmax2, uint64_t *hits, float *y_last)
{
    int nN = nA + nB;
    float *invP = (float *)malloc(sizeof(float) * (size_t)nN * 3);
    float *u0 = invP + nN, *u1 = u0 + nN;
    uint8_t *part = (uint8_t *)malloc((size_t)nN);
    for (int n = 0; n < nN; ++n) invP[n] = 1.f / orc_pdf_f32(targets[n].kind, targets[n].p0, targets[n].p1, x[n]);
    float sum = 0.0f, err = 0.0f;
    for (int s = 0; s < samples; ++s) {
        const float *d = draws + (size_t)s * nN * 4;
        for (int n = 0; n < nN; ++n) orc_mh_step_f32(&targets[n], &x[n], &invP[n], step, d[2 * n], d[2 * n + 1], NULL);
        const float *a = d + 2 * nN;
        for (int n = 0; n < nN; ++n) { /* :50-61 */
            u0[n] = a[2 * n];
            u1[n] = a[2 * n + 1];
            u0[n] = 1 - 2 * u0[n];
            u0[n] = sqrtf(1 - u0[n] * u0[n]);
            u1[n] = 2 * u1[n];
        }
        float y = 1;
        uint64_t ncoll = 0;
        if (hits) memset(part, 0, (size_t)nN);
        for (int i = 0; i < nA; ++i) {
            for (int j = 0; j < nB; ++j) { /* :70-75 */
                float ux = x[i] * u0[i] * orc_cospif(u1[i]);
                float uy = x[i] * u0[i] * orc_sinpif(u1[i]);
                float sx = x[j + nA] * u0[j + nA] * orc_cospif(u1[j + nA]);
                float sy = x[j + nA] * u0[j + nA] * orc_sinpif(u1[j + nA]);
                float dx = b + ux - sx, dy = uy - sy;
                y *= (1 - orc_GammaNN(dx, dy, sigmaNN, beta));
                if (hits && dx * dx + dy * dy < dmax2) {
                    ncoll += 1;
                    part[i] = 1;
                    part[j + nA] = 1;
                }
            }
        }
        if (hits) {
            hits[0] += ncoll;
            hits[1] += ncoll > 0;
            for (int n = 0; n < nN; ++n) hits[2] += part[n];
        }
        if (y_last) *y_last = y;
        y -= err; /* :81-85 */
        float t = sum + y;
        err = (t - sum) - y;
        sum = t;
    }
    free(invP);
    free(part);
    return (sum - err) / samples; /* :88 */
}

/* a15: xsection/src/main.cu:253-258 and :384-388.  run_means[R] ->
 * mean, stderr = sqrt(popvar / R); rms = sqrt(mean), rms_err = stderr/(2 rms). */
ORC_API void orc_run_estimator_f32(const float *run_means, int R, float *mean, float *stderr_, float *rms, float *rms_err)
{
    float m = orc_average_f32(run_means, R);
    float se = sqrtf(orc_variance_f32(run_means, m, R) / R);
    *mean = m;
    *stderr_ = se;
    if (rms) *rms = sqrtf(m);
    if (rms_err) *rms_err = se / (2 * sqrtf(m));
}

/* ------------------------------------------------------------------ */
/* f1: xsection/tools/XSectionAnalysis.py:3-28.                          */
static double orc_trapz(const double *x, const double *y, int n)
{
    double s = 0;
    for (int i = 0; i + 1 < n; ++i) s += (y[i] * 0.5 + y[i + 1] * 0.5) * (x[i + 1] - x[i]);
    return s;
}

static double orc_trapz_unc(const double *x, const double *e, int n)
{
    double s = 0;
    for (int i = 0; i + 1 < n; ++i) {
        double t = sqrt(e[i] * e[i] * 0.5 + e[i + 1] * e[i + 1] * 0.5) * (x[i + 1] - x[i]) / 2;
        s += t * t;
    }
    return sqrt(s);
}

/* out[6] = sigma_tot, unc, sigma_el, unc, sigma_rxn, unc  (mb). */
ORC_API void orc_cross_sections(const double *b, const double *S, const double *Serr, int n, double *out)
{
    double *y = (double *)malloc(sizeof(double) * (size_t)n * 2);
    double *e = y + n;
    for (int i = 0; i < n; ++i) { y[i] = b[i] * (1 - S[i]); e[i] = b[i] * Serr[i]; }
    out[0] = 10 * 2 * M_PI * 2 * orc_trapz(b, y, n);
    out[1] = 10 * orc_trapz_unc(b, e, n);
    for (int i = 0; i < n; ++i) { y[i] = b[i] * (1 - S[i]) * (1 - S[i]); e[i] = b[i] * sqrt(2.0) * (1 - S[i]) * Serr[i]; }
    out[2] = 10 * 2 * M_PI * orc_trapz(b, y, n);
    out[3] = 10 * orc_trapz_unc(b, e, n);
    for (int i = 0; i < n; ++i) { y[i] = b[i] * (1 - S[i] * S[i]); e[i] = b[i] * sqrt(2.0) * S[i] * Serr[i]; }
    out[4] = 10 * 2 * M_PI * orc_trapz(b, y, n);
    out[5] = 10 * orc_trapz_unc(b, e, n);
    free(y);
}

/* ------------------------------------------------------------------ */
/* Correlated Gaussian "stress" target (BASELINE.json configs[4]; product spec, no reference
 * function): precision matrix symmetric tridiagonal (q_diag[dim], q_off[dim-1]).  One sweep =
 * GibbsStep (xsection/include/Metropolis.cuh:69-79) = dim sequential MetropolisHastingsSteps
 * (:40-52) on the conditional density of the coordinate, accepted in log space:
 *   d = step * normal;  h = Q_ii x_i + Q_i,i-1 x_i-1 + Q_i,i+1 x_i+1;
 *   take  <=>  log U <= -(d * (h + (0.5 Q_ii) d)).
 * draws: [sweep][coord][2] (normal, uniform) or NULL -> the library's packed Philox stream
 * (tag 3, coord field 0, proposal index n = sweep * dim + i takes the place of the step).
 * trace (optional) [sweeps][dim] positions after each sweep; bits (optional) [sweeps][dim];
 * min_margin (optional) = min |log U - delta| over all proposals.                        */
ORC_API uint64_t orc_mvn_chain(int dim, const float *q_diag, const float *q_off, float *x, float step, uint64_t sweeps,
                               const float *draws, uint64_t seed, uint64_t chain, uint64_t sweep_offset,
                               float *trace, uint8_t *bits, float *min_margin)
{
    uint64_t acc = 0;
    float mm = INFINITY;
    for (uint64_t s = 0; s < sweeps; ++s) {
        for (int i = 0; i < dim; ++i) {
            float normal, uniform;
            if (draws) {
                normal = draws[(s * dim + i) * 2];
                uniform = draws[(s * dim + i) * 2 + 1];
            } else {
                uint64_t n = (sweep_offset + s) * (uint64_t)dim + (uint64_t)i;
                uint32_t m[4];
                orc_pair_fields_tag(seed, chain, 0u, n, 3u, m);
                float u1 = orc_field_uniform(m[0]);
                float rad = sqrtf(-2.0f * logf(u1));
                double ang = 2.0 * M_PI * ((double)m[1] * 1.1920928955078125e-07);
                normal = (n & 1u) ? rad * (float)sin(ang) : rad * (float)cos(ang);
                uniform = orc_field_uniform((n & 1u) ? m[3] : m[2]);
            }
            float d = step * normal;
            float qd = q_diag[i];
            float ol = i > 0 ? q_off[i - 1] : 0.0f, orr = i + 1 < dim ? q_off[i] : 0.0f;
            float xm = i > 0 ? x[i - 1] : 0.0f, xp = i + 1 < dim ? x[i + 1] : 0.0f;
            float h = qd * x[i] + ol * xm + orr * xp;
            float t1 = h + (0.5f * qd) * d;
            float delta = -(d * t1);
            float lu = logf(uniform);
            int take = lu <= delta;
            float gap = fabsf(lu - delta);
            if (gap < mm) mm = gap;
            if (take) { x[i] = x[i] + d; acc += 1; }
            if (bits) bits[s * dim + i] = (uint8_t)take;
        }
        if (trace) memcpy(trace + s * dim, x, sizeof(float) * (size_t)dim);
    }
    if (min_margin) *min_margin = mm;
    return acc;
}

/* ------------------------------------------------------------------ */
/* CPU baseline helper for bench.py: n_chains independent FP32 chains of the
 * reference GPU step logic (a5/a7) with library Philox draws, nC coordinates
 * each, `sweeps` sweeps; accumulates sum x, sum x^2 and accepts like the
 * streaming-statistics kernel.  Single thread; bench.py spreads chain ranges
 * over processes.  Returns accepted count.                                */
ORC_API uint64_t orc_baseline_chains_f32(uint64_t chain0, uint64_t n_chains, int nC, int kind, float p0, float p1,
                                         float step, uint64_t seed, uint64_t step_offset, uint64_t sweeps,
                                         float x0, double *sum1, double *sum2)
{
    uint64_t acc = 0;
    double s1 = 0, s2 = 0;
    orc_target tg = {kind, p0, p1};
    for (uint64_t c = chain0; c < chain0 + n_chains; ++c) {
        for (int d = 0; d < nC; ++d) {
            float x = x0;
            float invP = 1.f / orc_pdf_f32(kind, p0, p1, x);
            uint64_t t = step_offset, tb = step_offset + sweeps;
            while (t < tb) {
                /* one group of 8 steps = 3 Philox blocks, generated once (as the kernel does) */
                uint64_t g = t >> 3;
                uint32_t m[4][4];
                orc_group_fields(seed, c, (uint32_t)d, g, 0u, m);
                for (uint32_t j = 0; j < 4; ++j) {
                    float u1 = orc_field_uniform(m[j][0]);
                    float rad = sqrtf(-2.0f * logf(u1));
                    float ang = 6.283185307179586f * ((float)m[j][1] * 1.1920928955078125e-07f);
                    float z[2] = {rad * cosf(ang), rad * sinf(ang)};
                    float u[2] = {orc_field_uniform(m[j][2]), orc_field_uniform(m[j][3])};
                    for (uint32_t h = 0; h < 2; ++h) {
                        uint64_t ts = (g << 3) + 2 * j + h;
                        if (ts < t || ts >= tb) continue;
                        uint8_t bit;
                        orc_mh_step_f32(&tg, &x, &invP, step, z[h], u[h], &bit);
                        acc += bit;
                        s1 += x;
                        s2 += (double)x * x;
                    }
                }
                t = (g + 1) << 3;
            }
        }
    }
    if (sum1) *sum1 = s1;
    if (sum2) *sum2 = s2;
    return acc;
}

/* ---------------------------------------------------------------------
 * Randomised chain start (the improvement main.cu:144-148 describes; not implemented in the
 * reference): the density tabulated at the midpoints of 1024 equal bins of [0, r_max], inverse CDF
 * with the first word of Philox block (id, coord, blk 0, tag 4), linear inside the bin.  The table
 * is summed sequentially in double here (the device scans it in float, in tree order), so the two
 * agree to rounding of the table, not bit for bit.                           */
ORC_API float orc_init_from_pdf(int kind, float p0, float p1, uint64_t seed, uint64_t id, uint32_t coord)
{
    enum { NB = 1024 };
    static double cdf[NB + 1];
    const float rmax = (kind == ORC_WS || kind == ORC_R2WS) ? p0 + 20.0f * p1 : 7.0f * sqrtf(p0);
    const float h = rmax / NB;
    uint32_t w[4];
    int lo = 0, hi = NB, k;
    double target, den, v;
    cdf[0] = 0.0;
    for (k = 0; k < NB; k++) cdf[k + 1] = cdf[k] + (double)orc_pdf_f32(kind, p0, p1, ((float)k + 0.5f) * h);
    orc_block(seed, id, coord, 0u, 4u, w);
    target = (double)orc_u01_f32(w[0]) * cdf[NB];
    while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        if (cdf[mid] <= target) lo = mid;
        else hi = mid;
    }
    den = cdf[lo + 1] - cdf[lo];
    v = den > 0.0 ? (target - cdf[lo]) / den : 0.5;
    if (v > 1.0) v = 1.0;
    return (float)(((double)lo + v) * (double)h);
}
