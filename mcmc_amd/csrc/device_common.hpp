// device_common.hpp -- Philox4x32-10 in registers, draw transforms, target
// densities, accept rules and wave/block reductions shared by the gfx950
// kernels.  Written for CDNA4 only (wave64, DPP row ops, v_exp/v_log/v_rcp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mcmc_hip.h"

// The parity ("precise") paths must round exactly like the written
// expressions (the oracle is built with -ffp-contract=off); fused operations
// are requested explicitly with __builtin_fmaf where the fast paths want them.
#pragma clang fp contract(off)

namespace mcmc {

// ---------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Replaces the curandStateXORWOW array
// (xsection/include/Metropolis.cuh:10-24, 48 B/thread in HBM + curand_init
// skip-ahead): the state is the counter, rebuilt from (chain, coord, step).
// The key is wave-uniform, so the key schedule lives in SGPRs.
struct u32x4 {
    uint32_t x, y, z, w;
};

// MCMC_PHILOX_ROUNDS: 10 (Philox4x32-10, the default and the parity path) or 7 (Philox4x32-7, the fewest rounds that
// pass BigCrush per Salmon et al.; `make philox7` builds it as a separate library, bench.py reports its rate beside
// the default's).  Both are pinned by Random123's known-answer vectors (tests/test_oracle_pinning.py).
#ifndef MCMC_PHILOX_ROUNDS
#define MCMC_PHILOX_ROUNDS 10
#endif
static_assert(MCMC_PHILOX_ROUNDS == 10 || MCMC_PHILOX_ROUNDS == 7, "Philox4x32 with 10 or 7 rounds");

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < MCMC_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        u32x4 n;
        // three-input xor in one instruction: gfx950's v_bitop3_b32 with truth table 0x96 (a ^ b ^ c) issues
        // at the rate of a plain v_xor_b32 (tools/ubench3.hip); hipcc does not form it from two xors by itself
        n.x = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c.y, k0, 0x96);
        n.y = (uint32_t)p1;
        n.z = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c.w, k1, 0x96);
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// Draw convention (DESIGN.md "RNG"): ctr = (id_lo, coord<<8 | id_hi8, blk_lo, tag<<24 | blk_hi24).
//   TAG_MH      packed MH draws of the chain kernels: 3 blocks per group of 8 steps (below)
//   TAG_ANGLE   angle uniforms of the S_AB kernel, blk = step >> 1
//   TAG_SAB_MH  MH draws of the S_AB kernel, blk = step >> 1
constexpr uint32_t TAG_MH = 0u, TAG_ANGLE = 1u, TAG_SAB_MH = 2u; // (3 = TAG_MVN, mvn_kernel.hpp)

__device__ __forceinline__ u32x4 draw_block(uint64_t seed, uint64_t id, uint32_t coord, uint64_t blk, uint32_t tag)
{
    u32x4 c;
    c.x = (uint32_t)id;
    c.y = (coord << 8) | (uint32_t)((id >> 32) & 0xFFu);
    c.z = (uint32_t)blk;
    c.w = (tag << 24) | (uint32_t)((blk >> 32) & 0xFFFFFFu);
    return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// The same block for a caller that draws many blocks of ONE (id, coord, tag) stream: whatever of rounds 1 and 2 does not
// depend on the low block word is worked out once.  Round 1 multiplies c.x = id_lo (fixed) and c.z = blk_lo (wave-uniform:
// the scalar unit's product); its output word n.z = hi(M0 c.x) ^ c.w ^ k1 does not contain blk_lo, so round 2's product
// M1 * n.z is fixed as well.  Per block that removes one v_mad_u64_u32 and four of the twenty three-input xors (three
// two-input ones remain): 17 + 19 VALU instructions instead of 18 + 20.  Same bits as draw_block (pure algebra).
struct PhiloxStream {
    uint32_t yk;  // c.y ^ k0                  round 1: n.x = hi(M1 blk_lo) ^ yk
    uint32_t xk2; // hi(M1 z2) ^ (k0 + W0)     round 2: n.x = xk2 ^ lo(M1 blk_lo)
    uint32_t y2;  // lo(M1 z2)                 round 2: n.y
    uint32_t wk2; // lo(M0 c.x) ^ (k1 + W1)    round 2: n.z = hi(M0 x1) ^ wk2
    uint32_t cw;  // the counter word c.w = tag << 24 | blk_hi24 this was made for
};

constexpr uint32_t PHILOX_M0 = 0xD2511F53u, PHILOX_M1 = 0xCD9E8D57u, PHILOX_W0 = 0x9E3779B9u, PHILOX_W1 = 0xBB67AE85u;

__device__ __forceinline__ uint32_t philox_cw(uint64_t blk, uint32_t tag) { return (tag << 24) | (uint32_t)((blk >> 32) & 0xFFFFFFu); }

__device__ __forceinline__ PhiloxStream philox_stream(uint64_t seed, uint64_t id, uint32_t coord, uint32_t cw)
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint32_t cy = (coord << 8) | (uint32_t)((id >> 32) & 0xFFu);
    const uint64_t p0 = (uint64_t)PHILOX_M0 * (uint32_t)id;
    const uint32_t z2 = (uint32_t)(p0 >> 32) ^ cw ^ k1;
    const uint64_t q1 = (uint64_t)PHILOX_M1 * z2;
    PhiloxStream s;
    s.yk = cy ^ k0;
    s.xk2 = (uint32_t)(q1 >> 32) ^ (k0 + PHILOX_W0);
    s.y2 = (uint32_t)q1;
    s.wk2 = (uint32_t)p0 ^ (k1 + PHILOX_W1);
    s.cw = cw;
    return s;
}

__device__ __forceinline__ u32x4 draw_block_stream(const PhiloxStream &s, uint64_t seed, uint32_t blk_lo)
{
    uint32_t k0 = (uint32_t)seed + 2u * PHILOX_W0, k1 = (uint32_t)(seed >> 32) + 2u * PHILOX_W1;
    const uint64_t p1 = (uint64_t)PHILOX_M1 * blk_lo;               // wave-uniform
    const uint32_t x1 = (uint32_t)(p1 >> 32) ^ s.yk;                 // round 1, word x
    const uint64_t q0 = (uint64_t)PHILOX_M0 * x1;                    // round 2
    u32x4 c;
    c.x = s.xk2 ^ (uint32_t)p1;
    c.y = s.y2;
    c.z = (uint32_t)(q0 >> 32) ^ s.wk2;
    c.w = (uint32_t)q0;
#pragma unroll
    for (int r = 2; r < MCMC_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c.x;
        const uint64_t pz = (uint64_t)PHILOX_M1 * c.z;
        u32x4 n;
        n.x = __builtin_amdgcn_bitop3_b32((uint32_t)(pz >> 32), c.y, k0, 0x96);
        n.y = (uint32_t)pz;
        n.z = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c.w, k1, 0x96);
        n.w = (uint32_t)p0;
        c = n;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    return c;
}

// (0,1] like curand_uniform (Metropolis.cuh:47 relies on U != 0).
__device__ __forceinline__ float u01(uint32_t r)
{
    return __builtin_fmaf((float)r, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

// ---- packed MH draws -------------------------------------------------
// A group of 8 steps (4 step pairs) costs 3 Philox blocks instead of 4: every word gives a 23-bit field from its LOW
// 23 bits -- ONE v_bitop3_b32, (w & 0x7FFFFF) | 0x3F800000, turns it into a float in [1,2) (3.1 issue cycles; the
// v_alignbit_b32 that took the TOP 23 bits before cost 4.1) -- and the top bytes of the 12 words make the four fields of
// the fourth pair: field j = top bytes of words 0, 1, 2 of block j (24 bits, the low 23 kept), field 3 = top bytes of
// word 3 of blocks 0, 1, 2.  Two v_perm_b32 + one v_bitop3_b32 per spare field.  pair_words returns the fields of a
// pair as the bit patterns of those floats.
__device__ __forceinline__ uint32_t field12(uint32_t w)
{
    return __builtin_amdgcn_bitop3_b32(w, 0x007FFFFFu, 0x3F800000u, 0xEA); // (a & b) | c
}

__device__ __forceinline__ float mant12(uint32_t field_bits) { return __builtin_bit_cast(float, field_bits); }

constexpr float MANT_OPEN = 0.99999994f; // 1 - 2^-24: mant12(f) - MANT_OPEN = (2m+1) 2^-24 in (0,1)

// spare fields of a group, filled block by block; c carries the top bytes of the blocks' fourth words
struct Spare {
    uint32_t f[3];
    uint32_t c;
};

template <int J>
__device__ __forceinline__ void spare_take(const u32x4 &w, Spare &sp)
{
    // v_perm_b32: selector bytes 0-3 pick from the second operand, 4-7 from the first, 0x0c = zero
    const uint32_t t = __builtin_amdgcn_perm(w.y, w.x, 0x0c0c0703u);       // w0.b3 | w1.b3 << 8
    sp.f[J] = field12(__builtin_amdgcn_perm(w.z, t, 0x0c070100u));         // ... | w2.b3 << 16
    if constexpr (J == 0) sp.c = w.w;
    else if constexpr (J == 1) sp.c = __builtin_amdgcn_perm(w.w, sp.c, 0x0c0c0703u);
    else sp.c = field12(__builtin_amdgcn_perm(w.w, sp.c, 0x0c070100u));
}

__device__ __forceinline__ u32x4 block_fields(const u32x4 &w)
{
    u32x4 r;
    r.x = field12(w.x);
    r.y = field12(w.y);
    r.z = field12(w.z);
    r.w = field12(w.w);
    return r;
}

__device__ __forceinline__ u32x4 spare_fields(const Spare &sp)
{
    u32x4 r;
    r.x = sp.f[0];
    r.y = sp.f[1];
    r.z = sp.f[2];
    r.w = sp.c;
    return r;
}

// Fields of step pair J (0..3) of group g; sp carries the spares across J = 0..2 -> 3.
template <int J>
__device__ __forceinline__ u32x4 pair_words_tag(uint64_t seed, uint64_t id, uint32_t coord, uint64_t g, uint32_t tag, Spare &sp)
{
    if constexpr (J < 3) {
        const u32x4 w = draw_block(seed, id, coord, 3 * g + J, tag);
        spare_take<J>(w, sp);
        return block_fields(w);
    } else {
        return spare_fields(sp);
    }
}

template <int J>
__device__ __forceinline__ u32x4 pair_words(uint64_t seed, uint64_t id, uint32_t coord, uint64_t g, Spare &sp)
{
    return pair_words_tag<J>(seed, id, coord, g, TAG_MH, sp);
}

// The same fields through a PhiloxStream kept by the caller for the (id, coord, tag) it iterates over.  Precondition
// (the caller tests it outside its hot loop): the group's blocks 3g .. 3g+2 have the counter word st.cw.
template <int J>
__device__ __forceinline__ u32x4 pair_words_stream(const PhiloxStream &st, uint64_t seed, uint64_t g, Spare &sp)
{
    if constexpr (J < 3) {
        const u32x4 w = draw_block_stream(st, seed, (uint32_t)(3 * g + J));
        spare_take<J>(w, sp);
        return block_fields(w);
    } else {
        return spare_fields(sp);
    }
}

// Draws of one step pair from its fields: normals (z0 even step, z1 odd step) and accept uniforms.
//   MATH FAST + SCALED: returns step * z (scale2 = -2 ln2 step^2 folded into the radius)
template <int MATH, bool SCALED>
__device__ __forceinline__ void pair_draws_f32(const u32x4 &w, float scale2, float &z0, float &z1, float &u0, float &u1)
{
    const float f1 = mant12(w.x), f2 = mant12(w.y);
    if constexpr (MATH == MCMC_MATH_FAST) {
        // -2 ln u = (-2 ln2) log2 u; v_sin/v_cos take revolutions and reduce the argument themselves
        const float rad = __builtin_amdgcn_sqrtf((SCALED ? scale2 : -1.3862943611198906f) * __builtin_amdgcn_logf(f1 - MANT_OPEN));
        z0 = rad * __builtin_amdgcn_cosf(f2);
        z1 = rad * __builtin_amdgcn_sinf(f2);
    } else {
        const float rad = sqrtf(-2.0f * logf(f1 - MANT_OPEN));
        float sn, cs;
        sincospif(2.0f * (f2 - 1.0f), &sn, &cs);
        z0 = rad * cs;
        z1 = rad * sn;
    }
    u0 = mant12(w.z) - MANT_OPEN;
    u1 = mant12(w.w) - MANT_OPEN;
}

// FP64 flavour (stepsize CPU rule): uniforms in [0,1) like std::uniform_real_distribution (random.hpp:10)
__device__ __forceinline__ void pair_draws_f64(const u32x4 &w, double &z0, double &z1, double &u0, double &u1)
{
    const double a = (double)(mant12(w.x) - MANT_OPEN);
    const double b = (double)(mant12(w.y) - 1.0f);
    const double rad = sqrt(-2.0 * log(a));
    double sn, cs;
    sincospi(2.0 * b, &sn, &cs);
    z0 = rad * cs;
    z1 = rad * sn;
    u0 = (double)(mant12(w.z) - 1.0f);
    u1 = (double)(mant12(w.w) - 1.0f);
}

// Box-Muller pair from two raw 32-bit words (S_AB kernel streams).  FAST uses the hardware
// log2/sqrt and v_sin/v_cos (argument in revolutions, so no 2*pi multiply or reduction).
template <int MATH>
__device__ __forceinline__ void box_muller(uint32_t r0, uint32_t r1, float &z_even, float &z_odd)
{
    const float u1 = u01(r0), u2 = u01(r1);
    if constexpr (MATH == MCMC_MATH_FAST) {
        // -2 ln u = -2 ln2 * log2 u
        const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
        z_even = rad * __builtin_amdgcn_cosf(u2);
        z_odd = rad * __builtin_amdgcn_sinf(u2);
    } else {
        const float rad = sqrtf(-2.0f * logf(u1));
        float s, c;
        sincospif(2.0f * u2, &s, &c);
        z_even = rad * c;
        z_odd = rad * s;
    }
}

// ---------------------------------------------------------------------
// Target densities (Distributions.cuh:7-28, metropolis.hpp:9-17) with the
// parameters pre-digested once per launch.
template <typename Real>
struct Target {
    int kind;
    Real p0, p1; // WS: c, a.  Gauss: variance
    // fast-path constants (float only)
    float k1, k0; // WS: exp2(k1*r + k0) == exp((r-c)/a); Gauss: exp2(k1*r*r)
};

template <typename Real>
__device__ __forceinline__ Target<Real> make_target(const mcmc_target &t)
{
    Target<Real> o;
    o.kind = t.kind;
    o.p0 = (Real)t.p0;
    o.p1 = (Real)t.p1;
    const float log2e = 1.4426950408889634f;
    if (t.kind == MCMC_TARGET_WS || t.kind == MCMC_TARGET_R2WS) {
        o.k1 = log2e / t.p1;
        o.k0 = -t.p0 * log2e / t.p1;
    } else {
        o.k1 = -log2e / (2.0f * t.p0);
        o.k0 = 0.0f;
    }
    return o;
}

// Precise form: the written expression of the reference, KIND is a template
// argument so the switch costs nothing in the loop.
template <int KIND>
__device__ __forceinline__ float pdf_precise(const Target<float> &t, float r)
{
    if (r < 0.0f) return 0.0f;
    if constexpr (KIND == MCMC_TARGET_WS) return 1 / (1 + expf((r - t.p0) / t.p1));
    if constexpr (KIND == MCMC_TARGET_R2WS) return r * r / (1 + expf((r - t.p0) / t.p1));
    if constexpr (KIND == MCMC_TARGET_GAUSS) return expf(-r * r / (2 * t.p0));
    return r * r * expf(-r * r / (2 * t.p0));
}

template <int KIND>
__device__ __forceinline__ double pdf_precise(const Target<double> &t, double r)
{
    if (r < 0.0) return 0.0;
    if constexpr (KIND == MCMC_TARGET_WS) return 1 / (1 + exp((r - t.p0) / t.p1));
    if constexpr (KIND == MCMC_TARGET_R2WS) return r * r / (1 + exp((r - t.p0) / t.p1));
    if constexpr (KIND == MCMC_TARGET_GAUSS) return exp(-r * r / (2 * t.p0));
    return r * r * exp(-r * r / (2 * t.p0));
}

// Fast form: one FMA feeds v_exp_f32 (2^x), v_rcp_f32 for the quotient.
template <int KIND>
__device__ __forceinline__ float pdf_fast(const Target<float> &t, float r)
{
    float p;
    if constexpr (KIND == MCMC_TARGET_WS) {
        p = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(r, t.k1, t.k0)));
    } else if constexpr (KIND == MCMC_TARGET_R2WS) {
        // r < 0 -> 0 through the r^2 factor: one v_max instead of compare + select
        const float rp = __builtin_fmaxf(r, 0.0f);
        return (rp * rp) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(r, t.k1, t.k0)));
    } else if constexpr (KIND == MCMC_TARGET_GAUSS) {
        p = __builtin_amdgcn_exp2f(r * r * t.k1);
    } else {
        const float rp = __builtin_fmaxf(r, 0.0f);
        const float r2 = rp * rp;
        return r2 * __builtin_amdgcn_exp2f(r2 * t.k1);
    }
    return r < 0.0f ? 0.0f : p;
}

// log density (natural log) for the log-space rule; -inf for r < 0 (and r == 0 for r^2 targets).
template <int KIND>
__device__ __forceinline__ float logpdf_fast(const Target<float> &t, float r)
{
    const float ln2 = 0.6931471805599453f;
    float lp;
    if constexpr (KIND == MCMC_TARGET_WS || KIND == MCMC_TARGET_R2WS) {
        // -ln(1 + e^z) = -ln2 * log2(1 + 2^(z log2e))
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(r, t.k1, t.k0));
        lp = -__builtin_amdgcn_logf(1.0f + e);
        if constexpr (KIND == MCMC_TARGET_R2WS) lp = __builtin_fmaf(2.0f, __builtin_amdgcn_logf(r), lp);
        lp *= ln2;
    } else {
        lp = r * r * t.k1 * ln2; // -r^2/(2v)
        if constexpr (KIND == MCMC_TARGET_R2GAUSS) lp = __builtin_fmaf(2.0f * ln2, __builtin_amdgcn_logf(r), lp);
    }
    return r < 0.0f ? -__builtin_inff() : lp;
}

template <int KIND, int MATH>
__device__ __forceinline__ float pdf_eval(const Target<float> &t, float r)
{
    if constexpr (MATH == MCMC_MATH_FAST) return pdf_fast<KIND>(t, r);
    else return pdf_precise<KIND>(t, r);
}

// ---------------------------------------------------------------------
// One MH proposal in FP32 (a5, Metropolis.cuh:40-52).  `aux` (and `aux2`) are
// the cached per-chain quantities of the rule:
//   RECIP : aux = 1/p                       (the reference's cached reciprocal)
//   NODIV : aux = p
//   LOG   : aux = log p
//   CROSS : the density is kept as a quotient and the test p' >= U p is cross-multiplied, so no
//           reciprocal is evaluated at all:
//             WS      p = 1/d        aux = d = 1 + e^((r-c)/a)         d >= U d'
//             R2WS    p = n/(1+e)    aux = e, aux2 = n = r^2           n' (1 + e) >= (U n) (1 + e')
//             GAUSS   p = 2^g        aux = g = -r^2 log2e/(2v)         2^(g'-g) >= U
//             R2GAUSS p = n 2^g      aux = g, aux2 = n = r^2           n' 2^(g'-g) >= U n
template <int KIND, int RULE, int MATH>
__device__ __forceinline__ void mh_aux_init(const Target<float> &t, float x, float &aux, float &aux2)
{
    aux2 = 0.0f;
    if constexpr (RULE == MCMC_ACCEPT_RECIP) {
        const float p = pdf_eval<KIND, MATH>(t, x);
        if constexpr (MATH == MCMC_MATH_FAST) aux = __builtin_amdgcn_rcpf(p);
        else aux = 1.f / p; // Metropolis.cuh:64
    } else if constexpr (RULE == MCMC_ACCEPT_NODIV) {
        aux = pdf_eval<KIND, MATH>(t, x);
    } else if constexpr (RULE == MCMC_ACCEPT_LOG) {
        aux = logpdf_fast<KIND>(t, x);
    } else {
        const float inf = __builtin_inff();
        const float rp = __builtin_fmaxf(x, 0.0f);
        if constexpr (KIND == MCMC_TARGET_WS || KIND == MCMC_TARGET_R2WS) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(x, t.k1, t.k0));
            aux = KIND == MCMC_TARGET_R2WS ? e : (x < 0.0f ? inf : 1.0f + e);
            aux2 = rp * rp;
        } else {
            const float g = x * x * t.k1;
            aux = (KIND == MCMC_TARGET_GAUSS && x < 0.0f) ? -inf : g;
            aux2 = rp * rp;
        }
    }
}

template <int KIND, int RULE, int MATH>
__device__ __forceinline__ float mh_aux_init(const Target<float> &t, float x)
{
    static_assert(RULE != MCMC_ACCEPT_CROSS, "the cross-multiplied rule carries two cached values");
    float a, b;
    mh_aux_init<KIND, RULE, MATH>(t, x, a, b);
    return a;
}

// ---------------------------------------------------------------------
// The accept test `lhs >= rhs` and the conditional moves `d = take ? s : d` as ONE sequence under an EXEC mask:
// v_cmpx narrows EXEC to the accepting lanes and plain v_mov_b32 (2 issue cycles each) writes only those.
// hipcc's own form is v_cmp (vcc) + one v_cndmask_b32_e32 per value, and a v_cndmask that reads VCC measures ~7.7
// issue cycles on gfx950 (tools/ubench_cal.hip on MI355X, profiles/r2_ubench_cal.txt: v_cmp + 3 v_cndmask = 27 SIMD
// cycles per wave, v_cmpx + 3 v_mov + the EXEC restore = 10).  In the kernels the gain is smaller than in those
// pure loops and depends on occupancy: measured A/B on MI355X, chain kernel 134.4 -> 129.3 cycles per proposal and the
// nucleon chains 1.12 -> 1.25 T proposals/s at 8 waves per SIMD; the stress kernel (2 waves per SIMD) LOST 4 % and
// keeps hipcc's form; v_cmp_e64 into an SGPR pair + v_cndmask_e64 was slower than both (136.4 cycles).  EXEC is saved and restored inside the block, so it also works under a partial EXEC (inactive lanes read
// as not accepting).  v_cmpx_e32 writes VCC as well on gfx9.  Returns the mask of accepting lanes.
// No DPP instruction may follow within 5 issue slots (VALU/SALU write of EXEC -> DPP hazard, not tracked through
// inline asm): the callers' next DPP ops are the reductions at the end of a tile / sample.
__device__ __forceinline__ uint64_t select3_ge(float lhs, float rhs, float &d0, float s0, float &d1, float s1, float &d2, float s2)
{
    uint64_t mask, saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_ge_f32_e32 %[l], %[r]\n\t"
                 "v_mov_b32_e32 %[d0], %[s0]\n\t"
                 "v_mov_b32_e32 %[d1], %[s1]\n\t"
                 "v_mov_b32_e32 %[d2], %[s2]\n\t"
                 "s_mov_b64 %[m], exec\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [m] "=s"(mask), [sv] "=&s"(saved)
                 : [l] "v"(lhs), [r] "v"(rhs), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2)
                 : "vcc");
    return mask;
}

__device__ __forceinline__ uint64_t select2_ge(float lhs, float rhs, float &d0, float s0, float &d1, float s1)
{
    uint64_t mask, saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_ge_f32_e32 %[l], %[r]\n\t"
                 "v_mov_b32_e32 %[d0], %[s0]\n\t"
                 "v_mov_b32_e32 %[d1], %[s1]\n\t"
                 "s_mov_b64 %[m], exec\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [d0] "+v"(d0), [d1] "+v"(d1), [m] "=s"(mask), [sv] "=&s"(saved)
                 : [l] "v"(lhs), [r] "v"(rhs), [s0] "v"(s0), [s1] "v"(s1)
                 : "vcc");
    return mask;
}

// same with a second condition `gt > 0` (the NODIV rule never moves onto p' == 0): two v_cmpx in a row AND their masks
__device__ __forceinline__ uint64_t select2_ge_pos(float lhs, float rhs, float gt, float &d0, float s0, float &d1, float s1)
{
    uint64_t mask, saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_ge_f32_e32 %[l], %[r]\n\t"
                 "v_cmpx_lt_f32_e32 0, %[g]\n\t"
                 "v_mov_b32_e32 %[d0], %[s0]\n\t"
                 "v_mov_b32_e32 %[d1], %[s1]\n\t"
                 "s_mov_b64 %[m], exec\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [d0] "+v"(d0), [d1] "+v"(d1), [m] "=s"(mask), [sv] "=&s"(saved)
                 : [l] "v"(lhs), [r] "v"(rhs), [g] "v"(gt), [s0] "v"(s0), [s1] "v"(s1)
                 : "vcc");
    return mask;
}

// dz = proposal displacement (step * normal).  Returns the wave's mask of accepting lanes.
template <int KIND, int RULE, int MATH>
__device__ __forceinline__ uint64_t mh_step_dz(const Target<float> &t, float &x, float &aux, float &aux2, float dz,
                                               float uniform)
{
    const float next = x + dz;
    if constexpr (MATH != MCMC_MATH_FAST) {
        // the parity path: the reference's written expressions, hipcc's own compare + select
        static_assert(RULE == MCMC_ACCEPT_RECIP, "precise math implements the reference's rule only");
        const float pn = pdf_eval<KIND, MATH>(t, next);
        const bool take = pn * aux >= uniform; // Metropolis.cuh:47
        const float inv = 1 / pn;              // Metropolis.cuh:50
        aux = take ? inv : aux;
        x = take ? next : x;
        return __builtin_amdgcn_ballot_w64(take);
    } else if constexpr (RULE == MCMC_ACCEPT_RECIP) {
        const float pn = pdf_eval<KIND, MATH>(t, next);
        return select2_ge(pn * aux, uniform, aux, __builtin_amdgcn_rcpf(pn), x, next); // Metropolis.cuh:47-50
    } else if constexpr (RULE == MCMC_ACCEPT_NODIV) {
        const float pn = pdf_eval<KIND, MATH>(t, next);
        // p' == 0 with p == 0 can only happen for a bad start; never move onto p' == 0
        return select2_ge_pos(pn, uniform * aux, pn, aux, pn, x, next);
    } else if constexpr (RULE == MCMC_ACCEPT_LOG) {
        const float lpn = logpdf_fast<KIND>(t, next);
        const float lu = 0.6931471805599453f * __builtin_amdgcn_logf(uniform);
        return select2_ge(lpn - aux, lu, aux, lpn, x, next); // log U <= log p' - log p
    } else {
        const float inf = __builtin_inff();
        if constexpr (KIND == MCMC_TARGET_WS) {
            float dn = 1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(next, t.k1, t.k0));
            dn = next < 0.0f ? inf : dn;       // p' = 0: inf on the right-hand side rejects
            return select2_ge(aux, uniform * dn, aux, dn, x, next);
        } else if constexpr (KIND == MCMC_TARGET_R2WS) {
            // aux = e = d - 1: n' d >= (U n) d' as n' e + n' >= (U n) e' + (U n), two FMAs and no `1 + e'`
            const float en = __builtin_amdgcn_exp2f(__builtin_fmaf(next, t.k1, t.k0));
            // n' = r |r| (the |.| is a source modifier): r^2 for r >= 0; for r < 0 a negative left-hand side,
            // which fails against the non-negative right-hand side as n' = 0 would -- no v_max_f32 (4 cycles)
            const float nn = next * __builtin_fabsf(next);
            const float un = uniform * aux2;
            return select3_ge(__builtin_fmaf(nn, aux, nn), __builtin_fmaf(un, en, un), aux, en, aux2, nn, x, next);
        } else if constexpr (KIND == MCMC_TARGET_GAUSS) {
            float gn = next * next * t.k1;
            gn = next < 0.0f ? -inf : gn;
            return select2_ge(__builtin_amdgcn_exp2f(gn - aux), uniform, aux, gn, x, next);
        } else {
            const float nn = next * __builtin_fabsf(next); // see R2WS
            const float gn = __builtin_fabsf(nn) * t.k1;
            return select3_ge(nn * __builtin_amdgcn_exp2f(gn - aux), uniform * aux2, aux, gn, aux2, nn, x, next);
        }
    }
}

template <int KIND, int RULE, int MATH>
__device__ __forceinline__ uint64_t mh_step_dz(const Target<float> &t, float &x, float &aux, float dz, float uniform)
{
    static_assert(RULE != MCMC_ACCEPT_CROSS, "the cross-multiplied rule carries two cached values");
    float unused = 0.0f;
    return mh_step_dz<KIND, RULE, MATH>(t, x, aux, unused, dz, uniform);
}

template <int KIND, int RULE, int MATH>
__device__ __forceinline__ uint64_t mh_step(const Target<float> &t, float &x, float &aux, float step, float normal,
                                            float uniform)
{
    // Metropolis.cuh:43.  Deliberately mul + add (two roundings) in every mode: with a fused
    // multiply-add the positions drift from the reference arithmetic by an ulp per accepted step.
    return mh_step_dz<KIND, RULE, MATH>(t, x, aux, step * normal, uniform);
}

// ---------------------------------------------------------------------
// Sample write-back: one float per lane into a row out[0 .. row_bytes/4) whose base is wave-uniform.  A buffer
// store takes the 48-bit row base from SGPRs (the descriptor is rebuilt per row by the scalar unit) and a 32-bit
// byte offset from one VGPR that never changes, so a store is ONE vector-memory instruction with no VALU address
// arithmetic (a global_store needs a 64-bit VGPR address: v_lshl_add_u64 per store, and two registers kept per row in
// flight).  Streaming (nt) like the stores it replaces; lanes with lane_off_b >= row_bytes are dropped by the unit's
// range check.  0x00020000: DATA_FORMAT = 32 (the gfx9 raw-buffer word 3).
__device__ __forceinline__ void store_row_f32(const float *row, uint32_t row_bytes, uint32_t lane_off_b, float v)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(row), 0, (int)row_bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, (int)lane_off_b, 0, 2 /* nt */);
}

// Two samples per lane (consecutive steps of one chain) as ONE 64-bit store into a row of pairs
// out[0 .. row_bytes/8)[2]: half the vector-memory instructions of store_row_f32 for the same bytes (the
// "MCMC_SAMPLES_PAIRS" layout of mcmc_hip.h).  lane_off_b = 8 * chain.
__device__ __forceinline__ void store_row_f32x2(const float *row, uint32_t row_bytes, uint32_t lane_off_b, float v0, float v1)
{
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(row), 0, (int)row_bytes, 0x00020000);
    const u32x2 v = {__builtin_bit_cast(unsigned int, v0), __builtin_bit_cast(unsigned int, v1)};
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)lane_off_b, 0, 2 /* nt */);
}

// ---------------------------------------------------------------------
// Wavefront (64 lanes) reductions with DPP row operations, then the two
// cross-row steps through the permute network; block reductions through LDS.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    const long long r = ((long long)hi << 32) | (long long)(unsigned int)lo;
    return __builtin_bit_cast(double, r);
}

constexpr int DPP_QUAD_XOR1 = 0xB1;    // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;    // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

constexpr int DPP_ROW_BCAST15 = 0x142; // lane 15 of each row -> the next row (gfx9 wavefront broadcast)
constexpr int DPP_ROW_BCAST31 = 0x143; // lane 31 -> rows 2 and 3

// v from the broadcasting lane in the rows of ROW_MASK, +0.0 elsewhere
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_rows_f64(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}

// Sum over all 64 lanes; every lane receives the total.  The two cross-row steps are the wavefront-broadcast DPP
// controls + a v_readlane_b32 pair of lane 63 instead of two ds_bpermute_b32 round trips per half (the 49-lag kernel
// does 147 of these per tile).  (r3 + r2) + (r1 + r0): the same bits as the butterfly's (r0 + r1) + (r2 + r3).
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    v += dpp_f64<DPP_QUAD_XOR2>(v);
    v += dpp_f64<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f64<DPP_ROW_MIRROR>(v);
    v += dpp_rows_f64<DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_rows_f64<DPP_ROW_BCAST31, 0xC>(v);
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}

// Whole-wave (G = 64) reductions end with the two gfx9 wavefront-broadcast DPP steps (row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3) and ONE v_readlane_b32 of lane 63 instead of two ds_bpermute_b32 round trips through
// the LDS queue (~100 cycles of latency each, paid eight times per culled sample and once per pair pass before).  Rows
// outside the row mask receive `neutral`, so their lanes keep their value; only lane 63 is read.  For commutative
// operations the value is bit-identical to the butterfly's: (r3 op r2) op (r1 op r0) against (r0 op r1) op (r2 op r3).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_rows_f32(float neutral, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, neutral), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xF, false));
}

__device__ __forceinline__ float lane63_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Product over the G lanes of an aligned group (G power of two <= 64).
template <int G>
__device__ __forceinline__ float group_prod(float v)
{
    if constexpr (G >= 2) v *= dpp_f32<DPP_QUAD_XOR1>(v);
    if constexpr (G >= 4) v *= dpp_f32<DPP_QUAD_XOR2>(v);
    if constexpr (G >= 8) v *= dpp_f32<DPP_ROW_HALF_MIRROR>(v);
    if constexpr (G >= 16) v *= dpp_f32<DPP_ROW_MIRROR>(v);
    if constexpr (G == 64) {
        v *= dpp_rows_f32<DPP_ROW_BCAST15, 0xA>(1.0f, v);
        v *= dpp_rows_f32<DPP_ROW_BCAST31, 0xC>(1.0f, v);
        return lane63_f32(v);
    } else {
        if constexpr (G >= 32) v *= __shfl_xor(v, 16, 64);
        return v;
    }
}

// Minimum / maximum over all 64 lanes; every lane receives the result.
__device__ __forceinline__ float wave_minf(float v)
{
    v = __builtin_fminf(v, dpp_f32<DPP_QUAD_XOR1>(v));
    v = __builtin_fminf(v, dpp_f32<DPP_QUAD_XOR2>(v));
    v = __builtin_fminf(v, dpp_f32<DPP_ROW_HALF_MIRROR>(v));
    v = __builtin_fminf(v, dpp_f32<DPP_ROW_MIRROR>(v));
    v = __builtin_fminf(v, dpp_rows_f32<DPP_ROW_BCAST15, 0xA>(v, v));
    v = __builtin_fminf(v, dpp_rows_f32<DPP_ROW_BCAST31, 0xC>(v, v));
    return lane63_f32(v);
}

__device__ __forceinline__ float wave_maxf(float v) { return -wave_minf(-v); }

template <int G>
__device__ __forceinline__ uint32_t group_sum_u32(uint32_t v)
{
    if constexpr (G >= 2) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_QUAD_XOR1, 0xF, 0xF, true);
    if constexpr (G >= 4) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_QUAD_XOR2, 0xF, 0xF, true);
    if constexpr (G >= 8) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_HALF_MIRROR, 0xF, 0xF, true);
    if constexpr (G >= 16) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_MIRROR, 0xF, 0xF, true);
    if constexpr (G == 64) {
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST15, 0xA, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST31, 0xC, 0xF, false);
        return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    } else {
        if constexpr (G >= 32) v += (uint32_t)__shfl_xor((int)v, 16, 64);
        return v;
    }
}

} // namespace mcmc
