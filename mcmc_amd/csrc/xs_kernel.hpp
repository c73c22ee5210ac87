// xs_kernel.hpp -- the xsection device path:
//   xs_chain_kernel : WarmupMetropolis (Metropolis.cuh:83-108) and
//                     MCIntegrate_R2 (IntegrateR2.cuh:7-61).  The Gibbs sweep of
//                     the reference is nN independent 1-D chains (Metropolis.cuh:
//                     75-79 shares only the RNG stream), so with a counter RNG
//                     every (event, nucleon) pair is its own lane.
//   sab_kernel      : MCIntegrate_S_AB (IntegrateSAB.cuh:19-97).  One event per
//                     group of G lanes; nucleon state and transverse positions
//                     live in LDS, the A x B product is tiled: each lane keeps
//                     up to JB nucleons of B in registers and sweeps the A
//                     positions broadcast from LDS.
//   row reductions  : device Average (Statistics.cuh:23-25) in FP64, fixed order.
#pragma once
#include <type_traits>

#include "device_common.hpp"

namespace mcmc {

constexpr int XS_BLOCK = 256;

struct XsParams {
    uint64_t T;            // events on this device (reference totalThreads)
    uint64_t seed, step_offset, event_offset;
    uint64_t inj_stride;
    uint32_t nA, nB;
    uint32_t samples;
    float stepsize;
    float b;               // impact parameter
    float G0;              // sigmaNN / (4 PI beta), reference float arithmetic
    float twoBeta;         // 2.f * beta
    float b_stored;        // impact parameter in the units of the stored transverse positions
    float pos_scale;       // FAST: sqrt(log2(e) / (2 beta)), so that Gamma = G0 * 2^(-d'^2); PRECISE: 1
    float hit_d2;          // black-disk dmax^2 in the same units
    float cull_d;          // > 0: pairs farther apart than this (same units) change nothing (see sab_cull); 0: off
    uint32_t ga_log2;      // the G lanes of an event form a (1 << ga_log2) x (G >> ga_log2) grid over A x B
    const mcmc_target *targets;
    float *prevSample;
    const float *injected;
    float *result;         // R2: [nN][T]; SAB: [T]; warm-up: unused
    uint32_t *hits;        // SAB: [3][T] per-event counters, or nullptr
};

static __global__ void init_sample_kernel(float *prevSample, float v, uint64_t total)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) prevSample[i] = v;
}

// ---------------------------------------------------------------------
// Randomised chain starts.  The reference starts every chain at 1.0f and notes that it would be
// better to "randomize the initial states of the chain ... by treating it as a discrete pdf, and
// sampling according to that" (main.cu:144-148).  This is that: the nucleon's density tabulated on
// INIT_BINS equal bins of [0, r_max], inverse CDF with one Philox uniform (tag 4), linear inside
// the bin.  grid = (tiles, nN); every block rebuilds its nucleon's table (4 density values per thread).
constexpr int INIT_BINS = 1024;
constexpr uint32_t TAG_INIT = 4u;

__device__ __forceinline__ float pdf_any(const Target<float> &t, float r)
{
    switch (t.kind) {
    case MCMC_TARGET_WS: return pdf_precise<MCMC_TARGET_WS>(t, r);
    case MCMC_TARGET_R2WS: return pdf_precise<MCMC_TARGET_R2WS>(t, r);
    case MCMC_TARGET_GAUSS: return pdf_precise<MCMC_TARGET_GAUSS>(t, r);
    default: return pdf_precise<MCMC_TARGET_R2GAUSS>(t, r);
    }
}

// upper end of the table: the density is below ~1e-7 of its bulk value beyond it
__host__ __device__ inline float init_r_max(const mcmc_target &t)
{
    if (t.kind == MCMC_TARGET_WS || t.kind == MCMC_TARGET_R2WS) return t.p0 + 20.0f * t.p1;
    return 7.0f * sqrtf(t.p0);
}

static __global__ __launch_bounds__(XS_BLOCK) void init_from_pdf_kernel(const mcmc_target *targets, float *prevSample, uint64_t T,
                                                                 uint64_t seed, uint64_t event_offset)
{
    static_assert(INIT_BINS == 4 * XS_BLOCK, "four bins per thread");
    __shared__ float cdf[INIT_BINS + 1];
    __shared__ float part[XS_BLOCK];
    const uint32_t n = blockIdx.y, tid = threadIdx.x;
    const mcmc_target mt = targets[n];
    const Target<float> tg = make_target<float>(mt);
    const float h = init_r_max(mt) / INIT_BINS;
    float w[4], run = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        run += pdf_any(tg, ((float)(tid * 4 + q) + 0.5f) * h);
        w[q] = run;
    }
    part[tid] = run;
    __syncthreads();
    for (uint32_t off = 1; off < XS_BLOCK; off <<= 1) { // inclusive scan of the per-thread totals
        const float v = tid >= off ? part[tid - off] : 0.0f;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const float base = tid ? part[tid - 1] : 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) cdf[tid * 4 + q + 1] = base + w[q];
    if (tid == 0) cdf[0] = 0.0f;
    __syncthreads();
    const float total = cdf[INIT_BINS];
    for (uint64_t tile = blockIdx.x; tile * XS_BLOCK < T; tile += gridDim.x) {
        const uint64_t e = tile * XS_BLOCK + tid;
        if (e >= T) continue;
        const float target = u01(draw_block(seed, event_offset + e, n, 0, TAG_INIT).x) * total;
        uint32_t lo = 0, hi = INIT_BINS; // cdf[lo] <= target (< cdf[hi] unless the uniform rounded to 1)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cdf[mid] <= target) lo = mid;
            else hi = mid;
        }
        const float den = cdf[lo + 1] - cdf[lo];
        const float v = den > 0.0f ? __builtin_fminf((target - cdf[lo]) / den, 1.0f) : 0.5f;
        prevSample[(uint64_t)n * T + e] = ((float)lo + v) * h;
    }
}

// ---------------------------------------------------------------------
// grid = (ceil(T/256), nN): blockIdx.y is the nucleon, so the target kind is
// block-uniform and selected once outside the sweep loop.
template <int KIND, int RULE, int MATH, bool INJ, bool R2>
__device__ __forceinline__ void xs_chain_body(const XsParams &p, const mcmc_target &mt, uint32_t n, uint64_t e, bool valid)
{
    const Target<float> tg = make_target<float>(mt);
    const uint32_t nN = p.nA + p.nB;
    const uint64_t idx = (uint64_t)n * p.T + e;
    float x = p.prevSample[idx];
    float aux, aux2;
    mh_aux_init<KIND, RULE, MATH>(tg, x, aux, aux2);
    float sum = 0.0f, err = 0.0f;
    // Philox draws in FAST math: the proposal width is folded into the Box-Muller radius (as chain_kernel)
    constexpr bool PRESCALED = !INJ && MATH == MCMC_MATH_FAST;
    const float scale2 = -1.3862943611198906f * p.stepsize * p.stepsize;
    const uint64_t ta = p.step_offset, tb = p.step_offset + p.samples;
    const uint64_t gid = p.event_offset + e;
    PhiloxStream pstream = {};
    if constexpr (!INJ) pstream = philox_stream(p.seed, gid, n, philox_cw(3 * (ta >> 3), TAG_MH));
    (void)pstream;

    auto one = [&](uint64_t t, float z, float u) __attribute__((always_inline)) {
        if constexpr (INJ) {
            const float *d = p.injected + e * p.inj_stride + ((t - ta) * nN + n) * 2;
            z = d[0];
            u = d[1];
        }
        // Metropolis.cuh:43: mul + add, two roundings (PRESCALED: z already is stepsize * normal)
        mh_step_dz<KIND, RULE, MATH>(tg, x, aux, aux2, PRESCALED ? z : p.stepsize * z, u);
        if constexpr (R2) { // IntegrateR2.cuh:38-46
            float y = x * x;
            y -= err;
            const float tt = sum + y;
            err = (tt - sum) - y;
            sum = tt;
        }
    };

    // groups of 8 steps = 4 step pairs from 3 Philox blocks (packed draws, device_common.hpp)
    auto do_group = [&](auto full_tag, uint64_t g) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        Spare sp = {};
        auto body = [&](auto j_tag) __attribute__((always_inline)) {
            constexpr int J = decltype(j_tag)::value;
            float z0 = 0, z1 = 0, u0 = 0, u1 = 0;
            if constexpr (!INJ) {
                // full groups of the fast loop: the chain's precomputed Philox rounds (device_common.hpp, PhiloxStream)
                const u32x4 w = FULL ? pair_words_stream<J>(pstream, p.seed, g, sp) : pair_words<J>(p.seed, gid, n, g, sp);
                pair_draws_f32<MATH, PRESCALED>(w, scale2, z0, z1, u0, u1);
            }
            const uint64_t t0 = (g << 3) + 2 * J;
            if (FULL || (t0 >= ta && t0 < tb)) one(t0, z0, u0);
            if (FULL || (t0 + 1 >= ta && t0 + 1 < tb)) one(t0 + 1, z1, u1);
        };
        body(std::integral_constant<int, 0>{});
        body(std::integral_constant<int, 1>{});
        body(std::integral_constant<int, 2>{});
        body(std::integral_constant<int, 3>{});
    };
    if (p.samples > 0) {
        uint64_t g = ta >> 3;
        const uint64_t g_end = (tb + 7) >> 3, g_full_end = tb >> 3;
        if ((ta & 7) != 0 || (g << 3) + 8 > tb) {
            do_group(std::false_type{}, g);
            ++g;
        }
        // the stream holds while the block index 3g + j stays below the next multiple of 2^32: in chunks of 2^20 groups
        while (g < g_full_end) {
            uint64_t stop = (g | 0xFFFFFu) + 1;
            if (stop > g_full_end) stop = g_full_end;
            const uint32_t cw_a = philox_cw(3 * g, TAG_MH), cw_b = philox_cw(3 * stop - 1, TAG_MH);
            if (cw_a == cw_b) {
                if (cw_a != pstream.cw) pstream = philox_stream(p.seed, gid, n, cw_a);
                for (; g < stop; ++g) do_group(std::true_type{}, g);
            } else {
                for (; g < stop; ++g) do_group(std::false_type{}, g); // range-tested steps, generic draws
            }
        }
        if (g < g_end) do_group(std::false_type{}, g);
    }
    if (valid) {
        p.prevSample[idx] = x;
        if constexpr (R2) p.result[idx] = (sum - err) / p.samples; // IntegrateR2.cuh:51
    }
}

template <int RULE, int MATH, bool INJ, bool R2>
__global__ __launch_bounds__(XS_BLOCK) void xs_chain_kernel(const XsParams p)
{
    const uint32_t n = blockIdx.y;
    const uint64_t e0 = (uint64_t)blockIdx.x * XS_BLOCK + threadIdx.x;
    const bool valid = e0 < p.T;
    const uint64_t e = valid ? e0 : 0;
    const mcmc_target mt = p.targets[n];
    switch (mt.kind) {
    case MCMC_TARGET_WS: xs_chain_body<MCMC_TARGET_WS, RULE, MATH, INJ, R2>(p, mt, n, e, valid); break;
    case MCMC_TARGET_R2WS: xs_chain_body<MCMC_TARGET_R2WS, RULE, MATH, INJ, R2>(p, mt, n, e, valid); break;
    case MCMC_TARGET_GAUSS: xs_chain_body<MCMC_TARGET_GAUSS, RULE, MATH, INJ, R2>(p, mt, n, e, valid); break;
    default: xs_chain_body<MCMC_TARGET_R2GAUSS, RULE, MATH, INJ, R2>(p, mt, n, e, valid); break;
    }
}

// ---------------------------------------------------------------------
// Glauber S_AB kernel.
struct TgtLds {
    float p0, p1, k1, k0;
    int kind;
};

// sab_kernel's LDS header: the table of distinct densities, the run count and a byte index per nucleon (host + device)
constexpr int XS_TGT_MAX = 8;
// (the modes with hit counters are register-bound at 3 waves per SIMD whatever the table costs: they keep one entry per nucleon)
__host__ __device__ inline size_t xs_sab_header_bytes(size_t nN, bool per_nucleon)
{
    return per_nucleon ? ((sizeof(TgtLds) * nN + 15) / 16) * 16 : ((sizeof(TgtLds) * XS_TGT_MAX + 16 + nN + 15) / 16) * 16;
}

// NODIV for r^2 Woods-Saxon inside the S_AB kernels, FAST math: the test p' >= U p in the cross-multiplied form of the
// chain kernels' CROSS rule, with ONE cached value per nucleon (the LDS state is (x, aux)): aux = e = exp((r - c)/a),
// n = r |r| recomputed from r.  n' (1 + e) >= (U n)(1 + e') as two FMAs -- no reciprocal, no `1 + e'`, no v_max:
// 12 VALU instructions (37 issue cycles) instead of 14 (50).  A start at r < 0 (density 0) carries e = +inf: every
// proposal onto r' > 0 is accepted and none onto r' <= 0, as `p' * (1 / 0) >= U` does in the reference (Metropolis.cuh:47).
template <int MATH>
__device__ __forceinline__ constexpr bool sab_cross_form(int kind, int rule)
{
    return MATH == MCMC_MATH_FAST && rule == MCMC_ACCEPT_NODIV && kind == MCMC_TARGET_R2WS;
}

__device__ __forceinline__ void sab_mh_r2ws_cross(const Target<float> &t, float &x, float &e, float step, float z, float u)
{
    const float next = x + step * z; // Metropolis.cuh:43: mul + add
    const float en = __builtin_amdgcn_exp2f(__builtin_fmaf(next, t.k1, t.k0));
    const float nn = next * __builtin_fabsf(next);
    const float un = u * (x * __builtin_fabsf(x));
    select2_ge(__builtin_fmaf(nn, e, nn), __builtin_fmaf(un, en, un), e, en, x, next);
}

template <int KIND /* -1: per-nucleon kind */, int RULE, int MATH>
__device__ __forceinline__ void sab_mh(const TgtLds &tl, float &x, float &aux, float step, float z, float u)
{
    Target<float> tg;
    tg.kind = tl.kind;
    tg.p0 = tl.p0;
    tg.p1 = tl.p1;
    tg.k1 = tl.k1;
    tg.k0 = tl.k0;
    if constexpr (KIND >= 0) {
        if constexpr (sab_cross_form<MATH>(KIND, RULE)) sab_mh_r2ws_cross(tg, x, aux, step, z, u);
        else mh_step<KIND, RULE, MATH>(tg, x, aux, step, z, u);
    } else {
        switch (tl.kind) {
        case MCMC_TARGET_WS: mh_step<MCMC_TARGET_WS, RULE, MATH>(tg, x, aux, step, z, u); break;
        case MCMC_TARGET_R2WS:
            if constexpr (sab_cross_form<MATH>(MCMC_TARGET_R2WS, RULE)) sab_mh_r2ws_cross(tg, x, aux, step, z, u);
            else mh_step<MCMC_TARGET_R2WS, RULE, MATH>(tg, x, aux, step, z, u);
            break;
        case MCMC_TARGET_GAUSS: mh_step<MCMC_TARGET_GAUSS, RULE, MATH>(tg, x, aux, step, z, u); break;
        default: mh_step<MCMC_TARGET_R2GAUSS, RULE, MATH>(tg, x, aux, step, z, u); break;
        }
    }
}

template <int KIND, int RULE, int MATH>
__device__ __forceinline__ float sab_aux_init(const TgtLds &tl, float x)
{
    Target<float> tg;
    tg.kind = tl.kind;
    tg.p0 = tl.p0;
    tg.p1 = tl.p1;
    tg.k1 = tl.k1;
    tg.k0 = tl.k0;
    auto cross_e = [&]() __attribute__((always_inline)) {
        return x < 0.0f ? __builtin_inff() : __builtin_amdgcn_exp2f(__builtin_fmaf(x, tg.k1, tg.k0));
    };
    if constexpr (KIND >= 0) {
        if constexpr (sab_cross_form<MATH>(KIND, RULE)) return cross_e();
        else return mh_aux_init<KIND, RULE, MATH>(tg, x);
    } else {
        switch (tl.kind) {
        case MCMC_TARGET_WS: return mh_aux_init<MCMC_TARGET_WS, RULE, MATH>(tg, x);
        case MCMC_TARGET_R2WS:
            if constexpr (sab_cross_form<MATH>(MCMC_TARGET_R2WS, RULE)) return cross_e();
            else return mh_aux_init<MCMC_TARGET_R2WS, RULE, MATH>(tg, x);
        case MCMC_TARGET_GAUSS: return mh_aux_init<MCMC_TARGET_GAUSS, RULE, MATH>(tg, x);
        default: return mh_aux_init<MCMC_TARGET_R2GAUSS, RULE, MATH>(tg, x);
        }
    }
}

// Transverse position of a nucleon at radius r from two uniforms
// (IntegrateSAB.cuh:50-61,70-73): isotropic direction, z dropped.
template <int MATH>
__device__ __forceinline__ float2 transverse(float r, float u0, float u1)
{
    float ct = 1 - 2 * u0;
    float2 o;
    if constexpr (MATH == MCMC_MATH_FAST) {
        const float st = __builtin_amdgcn_sqrtf(__builtin_fmaf(-ct, ct, 1.0f));
        const float rs = r * st;
        o.x = rs * __builtin_amdgcn_cosf(u1); // cos(2 pi u1): v_cos takes revolutions
        o.y = rs * __builtin_amdgcn_sinf(u1);
    } else {
        const float st = sqrtf(1 - ct * ct);
        const float ph = 2 * u1;
        o.x = r * st * cospif(ph);
        o.y = r * st * sinpif(ph);
    }
    return o;
}

// Black-disk hit test `d2 < hit_d2` as a float 0 / 1 without a condition register: clamp(big * (hit_d2 - d2), 0, 1) in ONE
// v_fma_f32 with the clamp modifier.  big = 2^k is chosen from hit_d2's exponent so that the smallest positive difference
// of two floats below hit_d2 (half an ulp of hit_d2) still scales to >= 4: the result is exactly 1.0f for d2 < hit_d2 and
// exactly 0.0f otherwise (d2 = hit_d2 gives 0; a padded pair at d2 ~ 1e60 overflows to -inf and clamps to 0), i.e. the
// same decision bit for bit as the comparison.  hipcc's form of `count += (d2 < hit)` is v_cmp into an SGPR pair + v_addc:
// 16 SGPR pairs live in the unrolled tile made the kernel spill SGPRs; the float counters need none.
struct HitScale {
    float neg_big, hit_big;
};

__device__ __forceinline__ HitScale hit_scale(float hit_d2)
{
    HitScale h;
    const uint32_t e = (__builtin_bit_cast(uint32_t, hit_d2) >> 23) & 0xFFu; // biased exponent
    float big = 0x1p60f;
    if (e >= 30u && e <= 250u) big = __builtin_bit_cast(float, (280u - e) << 23); // 2^(26 - (e - 127))
    h.neg_big = -big;
    h.hit_big = hit_d2 > 0.0f ? hit_d2 * big : -1.0f; // no hit distance: the indicator is 0 for every d2 >= 0
    return h;
}

__device__ __forceinline__ float hit_indicator(float d2, const HitScale &h)
{
    float t;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(t) : "v"(d2), "s"(h.neg_big), "v"(h.hit_big));
    return t;
}

// The A x B pair tile of one lane.  The G lanes of an event form a GA x GB grid: lane (la, lb)
// owns the A nucleons i = la + GA*k (read from LDS, one ds_read_b64 per JC pairs, prefetched one
// iteration ahead) and keeps JC of its B nucleons j = lb + GB*(q0+q) in registers.  In FAST math
// the stored positions are pre-multiplied by sqrt(log2(e)/(2 beta)), so a pair costs
// 2 sub + mul + fma + v_exp_f32 + fma + mul.
// CHECK (one event per wavefront, PROD): every 32 A nucleons (8, 4 or 2 iterations by the number of lanes on A) the wave tests whether the product over the pairs done
// so far has underflowed to exactly 0 -- the remaining factors are finite and cannot change it.  With hit counters the
// loop goes on without the product (no exp); without them the sample's scan ends.  `dead` in: skip the product from the
// start; out: the product is 0.
template <int JC, int MATH, bool PROD, bool HITS, bool CHECK = false>
__device__ __forceinline__ void sab_pairs(const XsParams &p, const float2 *trE, uint8_t *partE, uint32_t nAe, uint32_t nBe,
                                          uint32_t la, uint32_t lb, uint32_t GA, uint32_t GB, uint32_t q0, float &prod,
                                          uint32_t &ncoll, bool &dead)
{
    // nAe / nBe: the A nucleons trE[0 .. nAe) and B nucleons trE[p.nA .. p.nA + nBe) that take part (all of them,
    // or what sab_cull kept)
    if constexpr (MATH == MCMC_MATH_FAST && JC % 2 == 0) {
        // Two B nucleons per register pair: every operation of the pair term is a packed-FP32 instruction on two pairs
        // (v_pk_add / mul / fma_f32) except v_exp_f32.  Same operations in the same order per pair as the scalar form
        // below: bit-identical results.  The A loop has a wave-uniform trip count (absent nucleons sit at -1e30: d^2 = inf,
        // factor exactly 1, never a hit), so that the whole wave can look at the product together.
        typedef float f2 __attribute__((ext_vector_type(2)));
        constexpr int H = JC / 2;
        f2 bx[H], by[H], pr[H], cq[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint32_t j = lb + (q0 + 2 * h + s) * GB;
                const bool ok = j < nBe;
                const float2 v = trE[p.nA + (ok ? j : 0)];
                bx[h][s] = ok ? v.x : 1.0e30f;
                by[h][s] = ok ? v.y : 0.0f;
            }
            pr[h] = f2{1.0f, 1.0f};
            cq[h] = f2{0.0f, 0.0f};
        }
        const HitScale hs = hit_scale(p.hit_d2);
        const f2 neg_big2 = {hs.neg_big, hs.neg_big}, hit_big2 = {hs.hit_big, hs.hit_big};
        const f2 one2 = {1.0f, 1.0f}, ng0 = {-p.G0, -p.G0};
        const float2 far = make_float2(-1.0e30f, 0.0f);
        const uint32_t KA = (nAe + GA - 1) >> p.ga_log2;
        bool alive = PROD && !(CHECK && dead); // wave-uniform
        // the wave looks at its product after every 32 A nucleons (8 iterations of 4, 4 of 8, 2 of 16 lanes on A: the plan for
        // Pb-Pb is GA = 16, whose 13 iterations a fixed period of 8 tested once, 128 A nucleons into the pass)
        const uint32_t check_mask = GA >= 16u ? 1u : GA >= 8u ? 3u : 7u;
        float2 a = la < nAe ? trE[la] : far;
        for (uint32_t k = 0; k < KA; ++k) {
            const uint32_t i = la + (k << p.ga_log2);
            const uint32_t in = i + GA;
            const bool more = in < nAe;
            float2 an = trE[more ? in : 0];
            if (!more) an = far;
            const f2 ax = {a.x, a.x}, ay = {a.y, a.y};
            float dmin = __builtin_inff();
            // the B slices against this A nucleon, with (the product is alive) or without the Gamma factor
            auto slices = [&](auto with_tag) __attribute__((always_inline)) {
                constexpr bool WITH = decltype(with_tag)::value;
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const f2 dx = ax - bx[h];
                    const f2 dy = ay - by[h];
                    const f2 d2 = __builtin_elementwise_fma(dy, dy, dx * dx);
                    if constexpr (WITH) {
                        const f2 ex = {__builtin_amdgcn_exp2f(-d2[0]), __builtin_amdgcn_exp2f(-d2[1])};
                        pr[h] *= __builtin_elementwise_fma(ng0, ex, one2);
                    }
                    if constexpr (HITS) {
                        f2 t;
                        asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(t) : "v"(d2), "v"(neg_big2), "v"(hit_big2));
                        cq[h] += t;
                        dmin = __builtin_fminf(__builtin_fminf(dmin, d2[0]), d2[1]); // one v_min3_f32
                    }
                }
            };
            if constexpr (PROD) {
                if (alive) {
                    slices(std::true_type{});
                    if constexpr (CHECK) {
                        if ((k & check_mask) == check_mask && k + 1 < KA) {
                            float w = prod;
#pragma unroll
                            for (int h = 0; h < H; ++h) w *= pr[h][0] * pr[h][1];
                            if (group_prod<64>(w) == 0.0f) {
                                dead = true;
                                alive = false;
                                if constexpr (!HITS) break; // nothing but the product was wanted
                            }
                        }
                    }
                } else {
                    slices(std::false_type{});
                }
            } else {
                slices(std::false_type{});
            }
            if constexpr (HITS) {
                if (dmin < p.hit_d2) partE[i] = 1; // benign race: every writer stores 1 (an absent nucleon never hits)
            }
            a = an;
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if constexpr (PROD) prod *= pr[h][s];
                if constexpr (HITS) {
                    ncoll += (uint32_t)cq[h][s];
                    if (cq[h][s] != 0.0f) partE[p.nA + lb + (q0 + 2 * h + s) * GB] = 1; // a padded j never collides
                }
            }
        }
        return;
    }
    const bool with_prod = PROD && !(CHECK && dead); // wave-uniform
    float bx[JC], by[JC], pr[JC];
    float cq[JC]; // HITS: collisions of B nucleon q (per-q counters double as the participant flags); exact integers < 2^24
#pragma unroll
    for (int q = 0; q < JC; ++q) {
        const uint32_t j = lb + (q0 + q) * GB;
        const bool ok = j < nBe;
        const float2 v = trE[p.nA + (ok ? j : 0)];
        bx[q] = ok ? v.x : 1.0e30f; // d^2 = inf -> Gamma = 0 -> factor exactly 1, never a hit
        by[q] = ok ? v.y : 0.0f;
        pr[q] = 1.0f;
        cq[q] = 0.0f;
    }
    const HitScale hs = hit_scale(p.hit_d2);
    uint32_t i = la;
    if (i < nAe) {
        float2 a = trE[i];
        for (;;) {
            const uint32_t in = i + GA;
            const bool more = in < nAe;
            const float2 an = trE[more ? in : i];
            float dmin = __builtin_inff(); // HITS: nearest of this lane's B nucleons to A nucleon i
#pragma unroll
            for (int q = 0; q < JC; ++q) {
                const float dx = a.x - bx[q]; // a.x already holds impactParameter + ux (IntegrateSAB.cuh:75)
                const float dy = a.y - by[q];
                float d2;
                if constexpr (MATH == MCMC_MATH_FAST) {
                    d2 = __builtin_fmaf(dy, dy, dx * dx);
                    if constexpr (PROD) {
                        if (with_prod) pr[q] *= __builtin_fmaf(-p.G0, __builtin_amdgcn_exp2f(-d2), 1.0f);
                    }
                } else {
                    d2 = dx * dx + dy * dy;
                    if constexpr (PROD) {
                        if (with_prod) pr[q] *= 1 - p.G0 * expf(-d2 / p.twoBeta); // IntegrateSAB.cuh:12,75
                    }
                }
                if constexpr (HITS) {
                    cq[q] += hit_indicator(d2, hs);
                    dmin = __builtin_fminf(dmin, d2);
                }
            }
            if constexpr (HITS) {
                if (dmin < p.hit_d2) partE[i] = 1; // benign race: every writer stores 1
            }
            if (!more) break;
            a = an;
            i = in;
        }
    }
#pragma unroll
    for (int q = 0; q < JC; ++q) {
        if constexpr (PROD) prod *= pr[q];
        if constexpr (HITS) {
            ncoll += (uint32_t)cq[q];
            if (cq[q] != 0.0f) partE[p.nA + lb + (q0 + q) * GB] = 1; // a padded j never collides
        }
    }
}

// Black-disk "does this event collide at all" test (MODE 4): only the running minimum of d^2 is kept
// (2 sub, mul, fma, half a v_min3 per pair; no exp, no counters, no participant marks).  The loop over
// the lane's A nucleons has a wave-uniform trip count (absent nucleons sit at -1e30), so that with one
// event per wavefront (G == 64) the first collision found anywhere in the wave ends the event's scan:
// returns true as soon as `ballot(dmin < hit_d2)` is non-zero.  For central heavy-ion events that is the
// first iteration.
template <int JC, int G, int MATH>
__device__ __forceinline__ bool sab_pairs_inel(const XsParams &p, const float2 *trE, uint32_t nAe, uint32_t nBe, uint32_t la,
                                               uint32_t lb, uint32_t GA, uint32_t GB, uint32_t q0, float &dmin)
{
    float bx[JC], by[JC];
#pragma unroll
    for (int q = 0; q < JC; ++q) {
        const uint32_t j = lb + (q0 + q) * GB;
        const bool ok = j < nBe;
        const float2 v = trE[p.nA + (ok ? j : 0)];
        bx[q] = ok ? v.x : 1.0e30f;
        by[q] = ok ? v.y : 0.0f;
    }
    const uint32_t KA = (nAe + GA - 1) >> p.ga_log2;
    const bool ok0 = la < nAe;
    float2 a = trE[ok0 ? la : 0];
    if (!ok0) a = make_float2(-1.0e30f, 0.0f);
    for (uint32_t k = 0; k < KA; ++k) {
        const uint32_t in = la + GA * (k + 1);
        const bool more = in < nAe;
        float2 an = trE[more ? in : 0];
        if (!more) an = make_float2(-1.0e30f, 0.0f);
#pragma unroll
        for (int q = 0; q < JC; ++q) {
            const float dx = a.x - bx[q];
            const float dy = a.y - by[q];
            if constexpr (MATH == MCMC_MATH_FAST) dmin = __builtin_fminf(dmin, __builtin_fmaf(dy, dy, dx * dx));
            else dmin = __builtin_fminf(dmin, dx * dx + dy * dy);
        }
        if constexpr (G == 64) {
            if (__builtin_amdgcn_ballot_w64(dmin < p.hit_d2) != 0) return true;
        }
        a = an;
    }
    return false;
}

// MODE bit 0: the S_AB product; bit 1: black-disk hit counters; MODE 4: inelastic flag only.
// (product + counters: three blocks per CU are what its LDS allows, so the register allocation is held to 168 VGPRs)
template <int G, int KIND, int RULE, int MATH, bool INJ, int MODE>
__global__ __launch_bounds__(XS_BLOCK, MODE == 3 ? 3 : 1) void sab_kernel(const XsParams p)
{
    constexpr bool PROD = (MODE & 1) != 0;
    constexpr bool HITS = (MODE & 2) != 0;
    constexpr bool INEL = (MODE & 4) != 0;
    static_assert(!INEL || MODE == 4, "the inelastic-only mode stands alone");
    constexpr int EPB = XS_BLOCK / G; // events per block
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t nN = p.nA + p.nB;
    // LDS carve-up (all in one array):
    //   TgtLds  tgtU[XS_TGT_MAX]  the distinct densities, in order of first appearance (runs of equal neighbours)
    //   uint32  nruns (+ pad)     number of runs found
    //   uint8   tidx[nN]          run index of every nucleon
    //   float2  st [EPB][nN]      (x, aux)
    //   float2  trA[EPB][nN]      transverse positions, even sweep of the pair
    //   float2  trB[EPB][nN]      odd sweep
    //   uint8   part[EPB][nN]     participant marks (HITS only)
    // A table entry per NUCLEON (20 B each, 8.3 KB for Pb-Pb) made the block 48.2 KB: three blocks per CU.  Nuclei have a
    // handful of species (protons / neutrons of A and B), so the table holds the distinct entries and a byte index per
    // nucleon: 40.5 KB, four blocks = 4 waves per SIMD.  More than XS_TGT_MAX runs: the densities are read from
    // p.targets in global memory at every use (slow, correct).
    TgtLds *tgtU = reinterpret_cast<TgtLds *>(smem);
    uint32_t *nruns = reinterpret_cast<uint32_t *>(smem + sizeof(TgtLds) * XS_TGT_MAX);
    uint8_t *tidx = smem + sizeof(TgtLds) * XS_TGT_MAX + 16;
    constexpr bool PER_NUCLEON = HITS; // one table entry per nucleon (see xs_sab_header_bytes)
    size_t off = xs_sab_header_bytes(nN, PER_NUCLEON);
    float2 *st = reinterpret_cast<float2 *>(smem + off);
    off += sizeof(float2) * (size_t)EPB * nN;
    float2 *trA = reinterpret_cast<float2 *>(smem + off);
    off += sizeof(float2) * (size_t)EPB * nN;
    float2 *trB = reinterpret_cast<float2 *>(smem + off);
    off += sizeof(float2) * (size_t)EPB * nN;
    uint8_t *part = reinterpret_cast<uint8_t *>(smem + off);

    const int tid = threadIdx.x;
    const int e_local = tid / G;
    const int lg = tid % G;
    const uint32_t GA = 1u << p.ga_log2, GB = (uint32_t)G >> p.ga_log2;
    const uint32_t la = (uint32_t)lg & (GA - 1), lb = (uint32_t)lg >> p.ga_log2;
    const uint64_t e0 = (uint64_t)blockIdx.x * EPB + e_local;
    const bool valid = e0 < p.T;
    const uint64_t e = valid ? e0 : 0;
    const uint64_t gid = p.event_offset + e;

    auto target_from_global = [&](uint32_t n) __attribute__((always_inline)) {
        const Target<float> t = make_target<float>(p.targets[n]);
        TgtLds tl;
        tl.p0 = t.p0;
        tl.p1 = t.p1;
        tl.k1 = t.k1;
        tl.k0 = t.k0;
        tl.kind = t.kind;
        return tl;
    };
    if constexpr (PER_NUCLEON) {
        for (uint32_t n = tid; n < nN; n += XS_BLOCK) tgtU[n] = target_from_global(n);
        __syncthreads();
    } else {
        // every nucleon's density once into scratch (the state / position arrays, not yet in use: 20 B <= 24 EPB B per
        // nucleon), then wave 0 numbers the runs of equal neighbours: ballot + prefix count, 64 nucleons a round
        TgtLds *tmp = reinterpret_cast<TgtLds *>(st);
        for (uint32_t n = tid; n < nN; n += XS_BLOCK) tmp[n] = target_from_global(n);
        __syncthreads();
        if (tid < 64) {
            uint32_t running = 0;
            for (uint32_t base = 0; base < nN; base += 64) {
                const uint32_t n = base + (uint32_t)tid;
                const bool ok = n < nN;
                const TgtLds a = tmp[ok ? n : 0];
                const TgtLds b = tmp[ok && n > 0 ? n - 1 : 0];
                const bool start = ok && (n == 0 || __builtin_bit_cast(uint32_t, a.p0) != __builtin_bit_cast(uint32_t, b.p0) ||
                                          __builtin_bit_cast(uint32_t, a.p1) != __builtin_bit_cast(uint32_t, b.p1) || a.kind != b.kind);
                const uint64_t m = __builtin_amdgcn_ballot_w64(start);
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const uint32_t idx = running + below + (start ? 1u : 0u) - 1u; // run of nucleon n (n = 0 starts run 0)
                if (ok) tidx[n] = (uint8_t)(idx < 255u ? idx : 255u);
                if (start && idx < (uint32_t)XS_TGT_MAX) tgtU[idx] = a;
                running += (uint32_t)__builtin_popcountll(m);
            }
            if (tid == 0) *nruns = running;
        }
        __syncthreads();
    }
    const bool tgt_table = PER_NUCLEON || *nruns <= (uint32_t)XS_TGT_MAX; // block-uniform
    auto target_of = [&](uint32_t n) __attribute__((always_inline)) {
        if constexpr (PER_NUCLEON) return tgtU[n];
        else return tgt_table ? tgtU[tidx[n]] : target_from_global(n);
    };
    float2 *stE = st + (size_t)e_local * nN;
    float2 *trAE = trA + (size_t)e_local * nN;
    float2 *trBE = trB + (size_t)e_local * nN;
    uint8_t *partE = part + (size_t)e_local * nN;
    for (uint32_t n = lg; n < nN; n += G) { // CopyLocalMHVars, Metropolis.cuh:61-66
        const float x = p.prevSample[(uint64_t)n * p.T + e];
        stE[n] = make_float2(x, sab_aux_init<KIND, RULE, MATH>(target_of(n), x));
        if constexpr (HITS) partE[n] = 0;
    }

    float sum = 0.0f, err = 0.0f;
    uint32_t h_coll = 0, h_inel = 0, h_part = 0;
    const uint64_t ta = p.step_offset, tb = p.step_offset + p.samples;
#ifdef MCMC_SAB_PROFILE
    // Diagnostic build (tools/gpu_sab_phases.sh): shader-clock stamps around the phases of a step pair; the per-event
    // totals REPLACE the kernel's outputs (hits[0..2] = MH/angle phase, culling + marks, pair sweeps; result = all).
    uint64_t pc_mh = 0, pc_cull = 0, pc_pair = 0;
    const uint64_t pc_begin = __builtin_amdgcn_s_memtime();
#define MCMC_PROF(acc, t0) acc += __builtin_amdgcn_s_memtime() - (t0)
#define MCMC_PROF_T(name) const uint64_t name = __builtin_amdgcn_s_memtime()
#else
#define MCMC_PROF(acc, t0) ((void)0)
#define MCMC_PROF_T(name) ((void)0)
#endif

    // After the target table above, every LDS word an event touches is private to the event's G lanes, and those lie
    // in ONE wavefront (G divides 64).  A wave's LDS operations execute in program order, so between one lane's
    // store and another lane's load of the same event a scheduling fence is all that is needed -- no workgroup
    // barrier: the four waves of a block no longer wait for each other's events (early exits, culled events).
    auto event_sync = []() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // Exact pair culling (one event per wavefront).  A pair farther apart than cull_d contributes the factor
    // exactly 1.0f (Gamma <= 2^-25 is absorbed by the rounding of 1 - Gamma, in the reference's arithmetic too)
    // and is never a black-disk hit, so an A nucleon outside the bounding box of B grown by cull_d -- and then a
    // B nucleon outside the grown box of the A nucleons kept -- can be dropped without changing any result bit
    // beyond the order of the multiplications.  The survivors are compacted in place (ballot prefix sums); the
    // pair tile then runs over nAe x nBe nucleons: peripheral collisions, where the zero-product exit never
    // fires, keep only the overlap strip.  Culling costs about 1000 issue cycles per sample, so it is skipped
    // where it cannot pay: for good once a sample keeps more than 7/8 of the pairs; in the product-only mode
    // while the previous sample of the event ended in an early zero-product exit; and in the inelastic-only
    // mode a short probe of the unculled nucleons comes first (central events collide there and are done).
    uint32_t nAe = p.nA, nBe = p.nB;
    bool cull_on = G == 64 && p.cull_d > 0.0f;
    bool cull_next = true;
    auto cull_side = [&](float2 *v0, uint32_t n0, float x0, float x1, float y0, float y1, float &kx0, float &kx1,
                         float &ky0, float &ky1) -> uint32_t {
        uint32_t cnt = 0;
        kx0 = ky0 = __builtin_inff();
        kx1 = ky1 = -__builtin_inff();
        for (uint32_t base = 0; base < n0; base += 64) {
            const uint32_t n = base + (uint32_t)lg;
            const bool ok = n < n0;
            const float2 v = v0[ok ? n : 0];
            const bool keep = ok && v.x >= x0 && v.x <= x1 && v.y >= y0 && v.y <= y1;
            const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
            const uint32_t pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (keep) {
                v0[pos] = v; // pos <= n, and every lane has read its element of this round already
                kx0 = __builtin_fminf(kx0, v.x);
                kx1 = __builtin_fmaxf(kx1, v.x);
                ky0 = __builtin_fminf(ky0, v.y);
                ky1 = __builtin_fmaxf(ky1, v.y);
            }
            cnt += (uint32_t)__builtin_popcountll(m);
        }
        return cnt;
    };
    auto sab_cull = [&](float2 *trE) {
        if constexpr (G == 64) {
            const float inf = __builtin_inff();
            float bx0 = inf, bx1 = -inf, by0 = inf, by1 = -inf;
            for (uint32_t n = p.nA + (uint32_t)lg; n < nN; n += 64) {
                const float2 v = trE[n];
                bx0 = __builtin_fminf(bx0, v.x);
                bx1 = __builtin_fmaxf(bx1, v.x);
                by0 = __builtin_fminf(by0, v.y);
                by1 = __builtin_fmaxf(by1, v.y);
            }
            const float cd = p.cull_d;
            float ax0, ax1, ay0, ay1, t0, t1, t2, t3;
            nAe = cull_side(trE, p.nA, wave_minf(bx0) - cd, wave_maxf(bx1) + cd, wave_minf(by0) - cd, wave_maxf(by1) + cd, ax0,
                            ax1, ay0, ay1);
            nBe = cull_side(trE + p.nA, p.nB, wave_minf(ax0) - cd, wave_maxf(ax1) + cd, wave_minf(ay0) - cd,
                            wave_maxf(ay1) + cd, t0, t1, t2, t3);
            if ((uint64_t)nAe * nBe * 8u > (uint64_t)p.nA * p.nB * 7u) cull_on = false;
        }
    };

    auto pair_sweep = [&](float2 *trE) {
        nAe = p.nA;
        nBe = p.nB;
        if constexpr (INEL) {
            float dmin = __builtin_inff();
            bool hit = false;
            if constexpr (G == 64) {
                if (cull_on) {
                    // probe: the first two A nucleons of every lane against its first 16 B nucleons
                    const uint32_t nAp = 2u * GA < p.nA ? 2u * GA : p.nA;
                    hit = sab_pairs_inel<16, G, MATH>(p, trE, nAp, p.nB, la, lb, GA, GB, 0u, dmin);
                    if (!hit) sab_cull(trE);
                }
            }
            const uint32_t JB = (nBe + GB - 1) / GB;
            for (uint32_t q0 = 0; q0 < JB && !hit;) {
                const uint32_t rem = JB - q0;
                if (rem >= 16) {
                    hit = sab_pairs_inel<16, G, MATH>(p, trE, nAe, nBe, la, lb, GA, GB, q0, dmin);
                    q0 += 16;
                } else if (rem >= 4) {
                    hit = sab_pairs_inel<4, G, MATH>(p, trE, nAe, nBe, la, lb, GA, GB, q0, dmin);
                    q0 += 4;
                } else {
                    hit = sab_pairs_inel<1, G, MATH>(p, trE, nAe, nBe, la, lb, GA, GB, q0, dmin);
                    q0 += 1;
                }
            }
            if constexpr (G != 64) hit = group_sum_u32<G>(dmin < p.hit_d2 ? 1u : 0u) != 0u;
            h_inel += hit ? 1u : 0u;
            return;
        }
        constexpr bool EXIT0 = PROD && !HITS && G == 64; // product only: a zero product ends the event sample
        constexpr bool DROP0 = PROD && HITS && G == 64;  // with hit counters: a zero product ends the exp work only
        MCMC_PROF_T(pc_c0);
        if (cull_on && (cull_next || !EXIT0)) sab_cull(trE);
        MCMC_PROF(pc_cull, pc_c0);
        MCMC_PROF_T(pc_p0);
        const uint32_t JB = (nBe + GB - 1) / GB;
        float prod = 1.0f;
        uint32_t ncoll = 0;
        // B slices of the lane in passes of 1, 2, 4, 8 (the binary digits of JB, smallest first), then 16s.
        // Product only, one event per wavefront: once the product over the pairs done so far has underflowed
        // to exactly 0 the remaining factors (all finite) cannot change it -- central heavy-ion events stop
        // after the small passes (Pb-Pb: 2 + 8 of 26 slices).  With hit counters every pair is still needed for
        // the counts, but the remaining passes run without the product (no exp).
        uint32_t q0 = 0;
        bool dead = false;
        auto pass = [&](auto jc_tag) __attribute__((always_inline)) {
            constexpr int JC = decltype(jc_tag)::value;
            if constexpr (DROP0) {
                // (the in-pass test of the product-only mode does not pay here: the product of a central event is gone after
                // the first one-slice pass, and the two loop bodies cost 19 VGPRs -- measured 117 against 126 M event samples/s)
                if (dead) sab_pairs<JC, MATH, false, true>(p, trE, partE, nAe, nBe, la, lb, GA, GB, q0, prod, ncoll, dead);
                else sab_pairs<JC, MATH, true, true>(p, trE, partE, nAe, nBe, la, lb, GA, GB, q0, prod, ncoll, dead);
            } else {
                if (dead) return;
                sab_pairs<JC, MATH, PROD, HITS, EXIT0>(p, trE, partE, nAe, nBe, la, lb, GA, GB, q0, prod, ncoll, dead);
            }
            q0 += JC;
            if constexpr (EXIT0 || DROP0) {
                if (!dead) dead = q0 < JB && group_prod<G>(prod) == 0.0f;
            }
        };
        if (JB & 1u) pass(std::integral_constant<int, 1>{});
        if (JB & 2u) pass(std::integral_constant<int, 2>{});
        if (JB & 4u) pass(std::integral_constant<int, 4>{});
        if (JB & 8u) pass(std::integral_constant<int, 8>{});
        while (q0 < JB && !(dead && !DROP0)) pass(std::integral_constant<int, 16>{});
        if (dead) prod = 0.0f;
        if constexpr (EXIT0) cull_next = !(dead && 2u * q0 <= JB);
        if constexpr (PROD) {
            float y = group_prod<G>(prod);
            // Kahan, IntegrateSAB.cuh:81-85 (every lane of the group carries the same sum)
            y -= err;
            const float tt = sum + y;
            err = (tt - sum) - y;
            sum = tt;
        }
        if constexpr (HITS) {
            ncoll = group_sum_u32<G>(ncoll);
            h_coll += ncoll;
            h_inel += ncoll > 0 ? 1u : 0u;
        }
        MCMC_PROF(pc_pair, pc_p0);
    };

    // participants of the sweep just done: count and clear the marks
    auto count_marks = [&]() {
        uint32_t c = 0;
        for (uint32_t n = lg; n < nN; n += G) {
            c += partE[n];
            partE[n] = 0;
        }
        h_part += group_sum_u32<G>(c);
    };

    // Inelastic-only mode, one event per wavefront, Philox draws: the angles of a sample are fresh draws, not
    // chain state, so they are only worked out for the nucleons a sweep actually looks at.  The MH loop leaves
    // the radii in the position buffers; the probe subset (the first 2 GA nucleons of A, the first 4 GB of B, 64 at most)
    // gets its angles (one Philox block per nucleon and step pair + sqrt/sin/cos) at once, the other nucleons
    // only if a probe of this step pair finds no collision.  Counter-based draws: the positions are bit-identical
    // to the eager evaluation, whichever nucleons are skipped.
    // An event whose probe misses once (peripheral: the probe costs more than it saves) goes back to the eager
    // evaluation for the rest of the launch.
    constexpr bool LAZY = INEL && G == 64 && !INJ;
    bool lazy_on = true;

    if (p.samples > 0) {
        for (uint64_t blk = ta >> 1; blk <= ((tb - 1) >> 1); ++blk) {
            const uint64_t te = blk << 1;
            const bool actA = te >= ta, actB = te + 1 < tb;
            bool lazy = false, needA = actA, needB = actB; // need*: the sweep is still to be done by pair_sweep
            if constexpr (LAZY) lazy = cull_on && lazy_on;
            MCMC_PROF_T(pc_m0);
            // the table index of the NEXT nucleon is fetched one iteration ahead (tidx -> tgtU is a dependent LDS read)
            uint32_t ti = (!PER_NUCLEON && tgt_table && (uint32_t)lg < nN) ? tidx[lg] : 0u;
            for (uint32_t n = lg; n < nN; n += G) {
                float2 s = stE[n];
                TgtLds tl;
                if constexpr (PER_NUCLEON) {
                    tl = tgtU[n];
                } else {
                    const uint32_t ti_next = (tgt_table && n + G < nN) ? tidx[n + G] : 0u;
                    tl = tgt_table ? tgtU[ti] : target_from_global(n);
                    ti = ti_next;
                }
                float z0 = 0, z1 = 0, u0 = 0, u1 = 0, a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                if constexpr (!INJ) {
                    const u32x4 r = draw_block(p.seed, gid, n, blk, TAG_SAB_MH);
                    box_muller<MATH>(r.x, r.y, z0, z1);
                    u0 = u01(r.z);
                    u1 = u01(r.w);
                    if (!lazy) {
                        const u32x4 q = draw_block(p.seed, gid, n, blk, TAG_ANGLE);
                        a0 = u01(q.x);
                        a1 = u01(q.y);
                        a2 = u01(q.z);
                        a3 = u01(q.w);
                    }
                }
                const float shift = n < p.nA ? p.b_stored : 0.0f;
                if (actA) {
                    if constexpr (INJ) {
                        const float *d = p.injected + e * p.inj_stride + (te - ta) * nN * 4;
                        z0 = d[2 * n];
                        u0 = d[2 * n + 1];
                        a0 = d[2 * nN + 2 * n];
                        a1 = d[2 * nN + 2 * n + 1];
                    }
                    sab_mh<KIND, RULE, MATH>(tl, s.x, s.y, p.stepsize, z0, u0);
                    const float rs = MATH == MCMC_MATH_FAST ? s.x * p.pos_scale : s.x;
                    if (lazy) {
                        trAE[n] = make_float2(rs, 0.0f);
                    } else {
                        float2 t = transverse<MATH>(rs, a0, a1);
                        if (n < p.nA) t.x = shift + t.x;
                        trAE[n] = t;
                    }
                }
                if (actB) {
                    if constexpr (INJ) {
                        const float *d = p.injected + e * p.inj_stride + (te + 1 - ta) * nN * 4;
                        z1 = d[2 * n];
                        u1 = d[2 * n + 1];
                        a2 = d[2 * nN + 2 * n];
                        a3 = d[2 * nN + 2 * n + 1];
                    }
                    sab_mh<KIND, RULE, MATH>(tl, s.x, s.y, p.stepsize, z1, u1);
                    const float rs = MATH == MCMC_MATH_FAST ? s.x * p.pos_scale : s.x;
                    if (lazy) {
                        trBE[n] = make_float2(rs, 0.0f);
                    } else {
                        float2 t = transverse<MATH>(rs, a2, a3);
                        if (n < p.nA) t.x = shift + t.x;
                        trBE[n] = t;
                    }
                }
                stE[n] = s;
            }
            event_sync();
            MCMC_PROF(pc_mh, pc_m0);
            MCMC_PROF_T(pc_z0);
            if constexpr (LAZY) {
                if (lazy) {
                    // at most 64 nucleons in all, so that their angles are one round of the wave
                    const uint32_t nAq = 2u * GA < 32u ? 2u * GA : 32u;
                    const uint32_t nAp = nAq < p.nA ? nAq : p.nA;
                    const uint32_t nBq = 4u * GB < 64u - nAp ? 4u * GB : 64u - nAp;
                    const uint32_t nBp = nBq < p.nB ? nBq : p.nB;
                    // radius -> transverse position of nucleon n, both steps of the pair (IntegrateSAB.cuh:50-61)
                    auto angles = [&](uint32_t n) __attribute__((always_inline)) {
                        const u32x4 q = draw_block(p.seed, gid, n, blk, TAG_ANGLE);
                        const float shift = n < p.nA ? p.b_stored : 0.0f;
                        if (actA) {
                            float2 t = transverse<MATH>(trAE[n].x, u01(q.x), u01(q.y));
                            if (n < p.nA) t.x = shift + t.x;
                            trAE[n] = t;
                        }
                        if (actB) {
                            float2 t = transverse<MATH>(trBE[n].x, u01(q.z), u01(q.w));
                            if (n < p.nA) t.x = shift + t.x;
                            trBE[n] = t;
                        }
                    };
                    for (uint32_t t = lg; t < nAp + nBp; t += 64) angles(t < nAp ? t : p.nA + (t - nAp));
                    __builtin_amdgcn_wave_barrier();
                    // probes first; what they leave undecided goes through pair_sweep on complete positions
                    float dmin = __builtin_inff();
                    if (actA && sab_pairs_inel<4, G, MATH>(p, trAE, nAp, nBp, la, lb, GA, GB, 0u, dmin)) {
                        h_inel += 1u;
                        needA = false;
                    }
                    dmin = __builtin_inff();
                    if (actB && sab_pairs_inel<4, G, MATH>(p, trBE, nAp, nBp, la, lb, GA, GB, 0u, dmin)) {
                        h_inel += 1u;
                        needB = false;
                    }
                    if (needA || needB) {
                        lazy_on = false;
                        for (uint32_t n = lg; n < nN; n += 64)
                            if (!(n < nAp || (n >= p.nA && n < p.nA + nBp))) angles(n);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            if constexpr (LAZY) MCMC_PROF(pc_cull, pc_z0);
            MCMC_PROF_T(pc_s0);
            if (needA) pair_sweep(trAE);
            if constexpr (INEL) MCMC_PROF(pc_pair, pc_s0);
            if constexpr (HITS) {
                MCMC_PROF_T(pc_k0);
                event_sync();
                if (actA) count_marks();
                event_sync();
                MCMC_PROF(pc_cull, pc_k0);
            }
            MCMC_PROF_T(pc_s1);
            if (needB) pair_sweep(trBE);
            if constexpr (INEL) MCMC_PROF(pc_pair, pc_s1);
            if constexpr (HITS) {
                MCMC_PROF_T(pc_k1);
                event_sync();
                if (actB) count_marks();
                MCMC_PROF(pc_cull, pc_k1);
            }
            event_sync();
        }
    }

    for (uint32_t n = lg; n < nN; n += G)
        if (valid) p.prevSample[(uint64_t)n * p.T + e] = stE[n].x;
    if (valid && lg == 0) {
        if constexpr (PROD) p.result[e] = (sum - err) / p.samples; // IntegrateSAB.cuh:88
        if constexpr (INEL) p.hits[e] = h_inel;
#ifdef MCMC_SAB_PROFILE
        if constexpr (INEL) { // one output only: MCMC_SAB_PROFILE = 1 MH phase, 2 lazy angles + probes, 3 pair sweeps, 4 everything
            const uint64_t pc_all = __builtin_amdgcn_s_memtime() - pc_begin;
            p.hits[e] = (uint32_t)(MCMC_SAB_PROFILE == 1 ? pc_mh : MCMC_SAB_PROFILE == 2 ? pc_cull : MCMC_SAB_PROFILE == 3 ? pc_pair : pc_all);
        }
#endif
        if constexpr (HITS) {
            p.hits[e] = h_coll;
            p.hits[p.T + e] = h_inel;
            p.hits[2 * p.T + e] = h_part;
#ifdef MCMC_SAB_PROFILE
            p.hits[e] = (uint32_t)pc_mh;
            p.hits[p.T + e] = (uint32_t)pc_cull;
            p.hits[2 * p.T + e] = (uint32_t)pc_pair;
            if constexpr (PROD) p.result[e] = (float)(__builtin_amdgcn_s_memtime() - pc_begin);
#endif
        }
    }
#undef MCMC_PROF
#undef MCMC_PROF_T
}

// ---------------------------------------------------------------------
// S_AB for light nuclei (NA, NB <= 4: the reference's own default He-4 + He-4, main.cu:38-41, and the
// hydrogen / helium isotopes): one event per LANE with every nucleon in registers, as the reference does
// (IntegrateSAB.cuh:30-78) -- no LDS, no barriers.  Same draws (tags 1 and 2, one Philox block per nucleon
// and step pair and tag), same MH arithmetic and the same Kahan accumulation as sab_kernel, so the chain
// states and the per-event means are bit-identical to the general kernel's (the pair product is multiplied in
// that kernel's lane-grid order, see `product` below).
// Product only, Philox draws, FAST math, r^2 Woods-Saxon densities; everything else takes sab_kernel.
template <int NA, int NB, int RULE>
__global__ __launch_bounds__(XS_BLOCK) void sab_small_kernel(const XsParams p)
{
    constexpr int NN = NA + NB;
    constexpr int KIND = MCMC_TARGET_R2WS, MATH = MCMC_MATH_FAST;
    const uint64_t e0 = (uint64_t)blockIdx.x * XS_BLOCK + threadIdx.x;
    const bool valid = e0 < p.T;
    const uint64_t e = valid ? e0 : p.T - 1; // lanes past the end replay the last event and store nothing
    const uint64_t gid = p.event_offset + e;

    TgtLds tgt[NN]; // wave-uniform: stays in scalar registers
    float x[NN], aux[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        const Target<float> t = make_target<float>(p.targets[n]);
        tgt[n].p0 = t.p0, tgt[n].p1 = t.p1, tgt[n].k1 = t.k1, tgt[n].k0 = t.k0, tgt[n].kind = t.kind;
        x[n] = p.prevSample[(uint64_t)n * p.T + e];
        aux[n] = sab_aux_init<KIND, RULE, MATH>(tgt[n], x[n]);
    }

    float sum = 0.0f, err = 0.0f;
    const uint64_t ta = p.step_offset, tb = p.step_offset + p.samples;
    if (p.samples > 0) {
        for (uint64_t blk = ta >> 1; blk <= ((tb - 1) >> 1); ++blk) {
            const uint64_t te = blk << 1;
            const bool actA = te >= ta, actB = te + 1 < tb;
            float2 trA[NN], trB[NN];
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                const u32x4 r = draw_block(p.seed, gid, (uint32_t)n, blk, TAG_SAB_MH);
                float z0, z1;
                box_muller<MATH>(r.x, r.y, z0, z1);
                const u32x4 q = draw_block(p.seed, gid, (uint32_t)n, blk, TAG_ANGLE);
                trA[n] = trB[n] = make_float2(0.0f, 0.0f);
                if (actA) {
                    sab_mh<KIND, RULE, MATH>(tgt[n], x[n], aux[n], p.stepsize, z0, u01(r.z));
                    trA[n] = transverse<MATH>(x[n] * p.pos_scale, u01(q.x), u01(q.y));
                    if (n < NA) trA[n].x = p.b_stored + trA[n].x;
                }
                if (actB) {
                    sab_mh<KIND, RULE, MATH>(tgt[n], x[n], aux[n], p.stepsize, z1, u01(r.w));
                    trB[n] = transverse<MATH>(x[n] * p.pos_scale, u01(q.z), u01(q.w));
                    if (n < NA) trB[n].x = p.b_stored + trB[n].x;
                }
            }
            // The pair product in the order of sab_kernel at G = 4 lanes per event (what the general kernel runs for
            // these nuclei): virtual lane l = (la, lb) of the GA x GB grid multiplies, for each of its B nucleons
            // j = lb + q GB in turn, the factors of its A nucleons i = la + k GA, and the four lane products combine
            // as (p0 p1)(p2 p3) (group_prod<4>).  Same factors, same order: the per-event means are bit-identical to
            // the general kernel's, so the choice of kernel (which depends on how many events a device owns) cannot
            // show in the results.
            auto product = [&](auto gl_tag, const float2 (&tr)[NN]) __attribute__((always_inline)) {
                constexpr int GL = decltype(gl_tag)::value, GA = 1 << GL, GB = 4 >> GL;
                float pl[4];
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    const int la = l & (GA - 1), lb = l >> GL;
                    pl[l] = 1.0f;
#pragma unroll
                    for (int j = lb; j < NB; j += GB) {
                        float prq = 1.0f;
#pragma unroll
                        for (int i = la; i < NA; i += GA) { // IntegrateSAB.cuh:66-78
                            const float dx = tr[i].x - tr[NA + j].x, dy = tr[i].y - tr[NA + j].y;
                            const float d2 = __builtin_fmaf(dy, dy, dx * dx);
                            prq *= __builtin_fmaf(-p.G0, __builtin_amdgcn_exp2f(-d2), 1.0f);
                        }
                        pl[l] *= prq;
                    }
                }
                float y = (pl[0] * pl[1]) * (pl[2] * pl[3]);
                y -= err; // Kahan, IntegrateSAB.cuh:81-85
                const float tt = sum + y;
                err = (tt - sum) - y;
                sum = tt;
            };
            auto product_gl = [&](const float2 (&tr)[NN]) __attribute__((always_inline)) {
                if (p.ga_log2 == 0) product(std::integral_constant<int, 0>{}, tr);
                else if (p.ga_log2 == 1) product(std::integral_constant<int, 1>{}, tr);
                else product(std::integral_constant<int, 2>{}, tr);
            };
            if (actA) product_gl(trA);
            if (actB) product_gl(trB);
        }
    }
    if (valid) {
#pragma unroll
        for (int n = 0; n < NN; ++n) p.prevSample[(uint64_t)n * p.T + e] = x[n];
        p.result[e] = (sum - err) / p.samples; // IntegrateSAB.cuh:88
    }
}

// ---------------------------------------------------------------------
// Row reductions: out[row] = scale * sum(x[row][:]) in FP64, two fixed-order stages.
constexpr int RR_CHUNKS = 64;

template <typename T>
__global__ __launch_bounds__(256) void row_partial_kernel(const T *x, uint64_t cols, double *partials)
{
    __shared__ double sw[4];
    const uint32_t row = blockIdx.y;
    const uint32_t chunk = blockIdx.x;
    const uint64_t per = (cols + RR_CHUNKS - 1) / RR_CHUNKS;
    const uint64_t c0 = (uint64_t)chunk * per;
    const uint64_t c1 = c0 + per < cols ? c0 + per : cols;
    double acc = 0.0;
    for (uint64_t c = c0 + threadIdx.x; c < c1; c += 256) acc += (double)x[(uint64_t)row * cols + c];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[(uint64_t)row * RR_CHUNKS + chunk] = ((sw[0] + sw[1]) + sw[2]) + sw[3];
}

static __global__ void row_final_kernel(const double *partials, double *out, uint32_t rows, double scale, int accumulate)
{
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    double tot = 0.0;
    for (int c = 0; c < RR_CHUNKS; ++c) tot += partials[(uint64_t)row * RR_CHUNKS + c];
    tot *= scale;
    out[row] = accumulate ? out[row] + tot : tot;
}

static __global__ void set_double_kernel(double *dst, double v) { *dst = v; }

// Sharding-invariant row sums (the multi-GPU reduction of main.cu:221-223,359): every element is rounded ONCE to a
// multiple of 2^-FIXED_FRAC (|x| < 2^14: an r^2 mean or an S_AB mean) and the integers are summed -- integer addition
// is associative, so the total does not depend on how the columns are split over blocks, launches or GPUs.  The
// 54-bit integer q = l2 2^40 + l1 2^20 + l0 (l0, l1 in [0, 2^20), l2 signed) travels as three limb sums; each stays
// below 2^52 for up to 2^32 columns in the whole job, so the limbs are exact as doubles and survive a FP64 SUM
// all-reduce.  out[row] = {sum l2, sum l1, sum l0, count of elements that were not finite or >= 2^14}.
constexpr int FIXED_FRAC = 40;
constexpr int FIXED_LIMB = 20;

static __global__ __launch_bounds__(256) void row_fixed_partial_kernel(const float *x, uint64_t cols, int64_t *partials)
{
    __shared__ int64_t sw[4][4];
    const uint32_t row = blockIdx.y;
    const uint32_t chunk = blockIdx.x;
    const uint64_t per = (cols + RR_CHUNKS - 1) / RR_CHUNKS;
    const uint64_t c0 = (uint64_t)chunk * per;
    const uint64_t c1 = c0 + per < cols ? c0 + per : cols;
    int64_t a2 = 0, a1 = 0, a0 = 0, bad = 0;
    for (uint64_t c = c0 + threadIdx.x; c < c1; c += 256) {
        const float v = x[(uint64_t)row * cols + c];
        if (!(__builtin_fabsf(v) < 16384.0f)) {
            ++bad;
            continue;
        }
        const int64_t q = __float2ll_rn(v * 0x1p40f); // the scaling is exact; one rounding, to 2^-40
        a0 += q & ((1ll << FIXED_LIMB) - 1);
        a1 += (q >> FIXED_LIMB) & ((1ll << FIXED_LIMB) - 1);
        a2 += q >> (2 * FIXED_LIMB);
    }
    int64_t v[4] = {a2, a1, a0, bad};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
        if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        partials[((uint64_t)row * RR_CHUNKS + chunk) * 4 + k] = sw[0][k] + sw[1][k] + sw[2][k] + sw[3][k];
    }
}

static __global__ void row_fixed_final_kernel(const int64_t *partials, double *out, uint32_t rows)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 4) return;
    const uint32_t row = i >> 2, k = i & 3;
    int64_t tot = 0;
    for (int c = 0; c < RR_CHUNKS; ++c) tot += partials[((uint64_t)row * RR_CHUNKS + c) * 4 + k];
    out[(uint64_t)row * 4 + k] = (double)tot; // |tot| < 2^52: exact
}

} // namespace mcmc
