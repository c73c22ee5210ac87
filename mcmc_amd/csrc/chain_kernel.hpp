// chain_kernel.hpp -- the many-chain Metropolis-Hastings kernel with streaming
// statistics: one 1-D chain per lane, state and
// RNG counter in registers, per-chain accumulators for sum, sum of squares,
// accepts and lag-k products, reduced wave (DPP) -> block (LDS) -> grid
// (fixed-order partials).  No per-step HBM traffic unless sample write-back
// or injected draws are requested.
#pragma once
#include "device_common.hpp"

namespace mcmc {

constexpr int CHAIN_BLOCK = 256;
constexpr int CHAIN_WAVES = CHAIN_BLOCK / 64;

struct ChainParams {
    uint64_t n_chains;
    uint64_t n_steps;
    uint64_t seed, step_offset, chain_offset;
    uint32_t n_coords, n_lags, n_tiles;
    int window_start;
    double step, center;
    mcmc_target target;
    void *state;
    double *partials; // [n_coords][gridDim.x][len]
    void *ring;
    void *samples;
    const void *inj_normals;
    const void *inj_uniforms;
    uint8_t *accept_bits;
};

__device__ __forceinline__ float wave_sum_f32(float v)
{
    v += dpp_f32<DPP_QUAD_XOR1>(v);
    v += dpp_f32<DPP_QUAD_XOR2>(v);
    v += dpp_f32<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f32<DPP_ROW_MIRROR>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Real = float : Metropolis.cuh:40-52 arithmetic (rules RECIP / NODIV / LOG)
// Real = double: metropolis.hpp:35-47 arithmetic (rule SHORTCIRCUIT)
template <typename Real, int KIND, int RULE, int MATH, int NLAG, bool INJ>
__global__ __launch_bounds__(CHAIN_BLOCK) void chain_kernel(const ChainParams p)
{
    constexpr int NL = NLAG > 0 ? NLAG : 1;
    __shared__ double sred[CHAIN_WAVES][3 + 3 * NL];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t coord = blockIdx.y;
    const int len = 3 + 3 * (int)p.n_lags;

    for (int i = tid; i < CHAIN_WAVES * (3 + 3 * NL); i += CHAIN_BLOCK) (&sred[0][0])[i] = 0.0;
    __syncthreads();

    const Target<Real> tg = make_target<Real>(p.target);
    const Real step = (Real)p.step;
    const Real center = (Real)p.center;
    const uint64_t t_begin = p.step_offset;
    const uint64_t t_end = p.step_offset + p.n_steps;
    const uint64_t nc = p.n_chains;

    for (uint32_t tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const uint64_t chain = (uint64_t)tile * CHAIN_BLOCK + tid;
        const bool valid = chain < nc;
        const uint64_t ch = valid ? chain : 0; // clamp so that every lane runs the same code on legal addresses
        const uint64_t gid = p.chain_offset + ch;
        Real *state = (Real *)p.state + (uint64_t)coord * nc + ch;

        Real x = *state;
        Real aux; // float: 1/p | p | log p ; double: p(x)
        if constexpr (sizeof(Real) == 4) aux = mh_aux_init<KIND, RULE, MATH>(tg, x);
        else aux = pdf_precise<KIND>(tg, x);

        Real hist[NL];
        float P[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            P[k] = 0.0f;
            hist[k] = (Real)0;
        }
        if (NLAG > 0 && !p.window_start && p.ring != nullptr) {
#pragma unroll
            for (int k = 0; k < NL; ++k)
                if (k < (int)p.n_lags) hist[k] = ((const Real *)p.ring)[((uint64_t)coord * p.n_lags + k) * nc + ch];
        }

        double S1 = 0.0, S2 = 0.0;
        float s1 = 0.0f, s2 = 0.0f;
        uint32_t nacc = 0;
        uint64_t cursor = 0; // injected uniforms consumed (rule SHORTCIRCUIT)
        (void)cursor;

        // One sweep of this chain: global step t.
        auto one_step = [&](uint64_t t, Real normal, Real uniform) {
            const uint64_t i = t - t_begin; // step within this launch
            const uint64_t slot = (i * p.n_coords + coord) * nc + ch;
            if constexpr (INJ) {
                normal = ((const Real *)p.inj_normals)[slot];
                if constexpr (RULE != MCMC_ACCEPT_SHORTCIRCUIT) uniform = ((const Real *)p.inj_uniforms)[slot];
            }
            bool take;
            if constexpr (sizeof(Real) == 4) {
                take = mh_step<KIND, RULE, MATH>(tg, x, aux, step, normal, uniform);
            } else {
                // metropolis.hpp:36-44
                const Real proposed = x + step * normal;
                const Real pp = pdf_precise<KIND>(tg, proposed);
                take = pp > aux;
                if (!take) {
                    if constexpr (INJ) {
                        uniform = ((const Real *)p.inj_uniforms)[(cursor * p.n_coords + coord) * nc + ch];
                        cursor += 1;
                    }
                    take = uniform < pp / aux;
                }
                if (take) {
                    x = proposed;
                    aux = pp;
                }
            }
            nacc += take ? 1u : 0u;
            const Real y = x - center;
            s1 += (float)y;
            s2 = __builtin_fmaf((float)y, (float)y, s2);
            if constexpr (NLAG > 0) {
#pragma unroll
                for (int k = 0; k < NL; ++k) P[k] = __builtin_fmaf((float)y, (float)hist[k], P[k]);
#pragma unroll
                for (int k = NL - 1; k > 0; --k) hist[k] = hist[k - 1];
                hist[0] = y;
            }
            if (p.samples != nullptr && valid) ((Real *)p.samples)[slot] = x;
            if (p.accept_bits != nullptr && valid) p.accept_bits[slot] = take ? 1 : 0;
        };

        auto run_steps = [&](uint64_t ta, uint64_t tb) {
            if (ta >= tb) return;
            for (uint64_t blk = ta >> 1; blk <= ((tb - 1) >> 1); ++blk) {
                Real z0 = 0, z1 = 0, u0 = 0, u1 = 0;
                if constexpr (!INJ) {
                    const u32x4 r = draw_block(p.seed, gid, coord, blk, 0u);
                    if constexpr (sizeof(Real) == 4) {
                        float a, b;
                        box_muller<MATH>(r.x, r.y, a, b);
                        z0 = a;
                        z1 = b;
                        u0 = u01(r.z);
                        u1 = u01(r.w);
                    } else {
                        double a, b;
                        box_muller_f64(r.x, r.y, a, b);
                        z0 = a;
                        z1 = b;
                        u0 = (double)r.z * 2.3283064365386963e-10;
                        u1 = (double)r.w * 2.3283064365386963e-10;
                    }
                }
                const uint64_t te = blk << 1;
                if (te >= ta && te < tb) one_step(te, z0, u0);
                if (te + 1 >= ta && te + 1 < tb) one_step(te + 1, z1, u1);
                if ((blk & 63u) == 63u) {
                    S1 += (double)s1;
                    S2 += (double)s2;
                    s1 = 0.0f;
                    s2 = 0.0f;
                }
            }
        };

        // Phase 1: the first NLAG steps of a window, after which the ring holds
        // exactly the window's first NLAG samples (the "head" sums A_i).
        uint64_t t_mid = t_begin;
        if constexpr (NLAG > 0) {
            if (p.window_start && p.n_steps >= (uint64_t)p.n_lags) {
                t_mid = t_begin + p.n_lags;
                run_steps(t_begin, t_mid);
#pragma unroll
                for (int k = 0; k < NL; ++k) {
                    if (k < (int)p.n_lags) {
                        // hist[k] = y_{n_lags-1-k}
                        const double w = wave_sum(valid ? (double)hist[k] : 0.0);
                        if (lane == 0) sred[wave][3 + p.n_lags + (p.n_lags - 1 - k)] += (double)w;
                    }
                }
            }
        }
        run_steps(t_mid, t_end);
        S1 += (double)s1;
        S2 += (double)s2;

        // clamped (invalid) lanes replayed chain 0 on a possibly stale read: they never store
        if (valid) *state = x;

        if (NLAG > 0 && p.ring != nullptr && valid) {
#pragma unroll
            for (int k = 0; k < NL; ++k)
                if (k < (int)p.n_lags) ((Real *)p.ring)[((uint64_t)coord * p.n_lags + k) * nc + ch] = hist[k];
        }

        // wave reductions -> per-wave LDS accumulators (only lane 0 of a wave touches its row)
        const double r1 = wave_sum(valid ? S1 : 0.0);
        const double r2 = wave_sum(valid ? S2 : 0.0);
        const double r3 = wave_sum(valid ? (double)nacc : 0.0);
        if (lane == 0) {
            sred[wave][0] += r1;
            sred[wave][1] += r2;
            sred[wave][2] += r3;
        }
        if constexpr (NLAG > 0) {
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                if (k < (int)p.n_lags) {
                    const double pk = wave_sum(valid ? (double)P[k] : 0.0);
                    const double zk = wave_sum(valid ? (double)hist[k] : 0.0);
                    if (lane == 0) {
                        sred[wave][3 + k] += (double)pk;
                        sred[wave][3 + 2 * p.n_lags + k] += (double)zk;
                    }
                }
            }
        }
    }

    __syncthreads();
    if (tid < len) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < CHAIN_WAVES; ++w) tot += sred[w][tid];
        p.partials[((uint64_t)coord * gridDim.x + blockIdx.x) * len + tid] = tot;
    }
}

// Fixed-order sum of the block partials: grid = n_coords, one thread per entry.
static __global__ void chain_finalize_kernel(const double *partials, double *stats, int len, int nblocks)
{
    const int coord = blockIdx.x;
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
        double tot = 0.0;
        for (int b = 0; b < nblocks; ++b) tot += partials[((uint64_t)coord * nblocks + b) * len + i];
        stats[(uint64_t)coord * len + i] = tot;
    }
}

template <typename Real, int KIND, int RULE, int MATH, int NLAG, bool INJ>
void launch_chain(const ChainParams &p, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL((chain_kernel<Real, KIND, RULE, MATH, NLAG, INJ>), grid, dim3(CHAIN_BLOCK), 0, s, p);
}

} // namespace mcmc
