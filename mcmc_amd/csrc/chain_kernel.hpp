// chain_kernel.hpp -- the many-chain Metropolis-Hastings kernel with streaming
// statistics: one 1-D chain per lane, state and
// RNG counter in registers, per-chain accumulators for sum, sum of squares,
// accepts and lag-k products, reduced wave (DPP) -> block (LDS) -> grid
// (fixed-order partials).  No per-step HBM traffic unless sample write-back
// or injected draws are requested.
#pragma once
#include <type_traits>
#include <utility>

#include "device_common.hpp"

namespace mcmc {

constexpr int CHAIN_BLOCK = 256;
constexpr int CHAIN_WAVES = CHAIN_BLOCK / 64;
constexpr int LAGPAD = 8; // steps per lag-window group = one draw group; the window is shifted once per group

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename F, int... Is>
__device__ __forceinline__ void expand_sequence(F &&f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}

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
    float lag_scale;  // chain_lag_kernel: power of two that maps |x - center| into f16's normal range
    int lag_mfma;     // FP32 with lags: 1 = chain_lag_kernel (lag products on the matrix cores), 0 = the VALU form below
    int samples_pairs; // samples in the pair layout [n_steps / 2][n_coords][n_chains][2] (STORE == 3)
    uint64_t *stamps; // STAMP instantiation only: [n_coords][gridDim.x][2] = shader cycles, 100 MHz ticks of the block's tile loop
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

// Real = float : Metropolis.cuh:40-52 arithmetic (rules RECIP / NODIV / LOG / CROSS)
// Real = double: metropolis.hpp:35-47 arithmetic (rule SHORTCIRCUIT)
// STORE: 0 = no write-back; 1 = sample and / or accept-flag write-back (which of the two is tested at run time);
//        2 = samples only, FP32 FAST-math Philox kernels without lags (the throughput case of the write-back path):
//            no run-time tests in the step, the row (step, coordinate) addressed by a wave-uniform base + the lane's
//            32-bit byte offset;
//        3 = as 2 in the pair layout samples[n_steps / 2][n_coords][n_chains][2] (option "MCMC_SAMPLES_PAIRS"): two
//            consecutive steps of a chain leave as one 64-bit store (step_offset and n_steps even)
// STAMP: diagnostic instantiation (mcmc_chain_clock_probe) -- the same body with one s_memtime / s_memrealtime pair
//        around the block's tile loop, written to a buffer of their own; no production launch executes a stamp
template <typename Real, int KIND, int RULE, int MATH, int NLAG, bool INJ, int STORE, bool STAMP = false>
#ifndef MCMC_LAG_BLOCKS_PER_CU
#define MCMC_LAG_BLOCKS_PER_CU 3
#endif
// FP32 with lags: 3 blocks per CU (168 VGPRs hold the 49 + 57 lag registers).  FP64 with lags keeps 49 double samples and
// 49 double accumulators beside the float ones: no register cap (one block per CU), or it spills ~700 registers.
__global__ __launch_bounds__(CHAIN_BLOCK, (NLAG > 0 && sizeof(Real) == 4) ? MCMC_LAG_BLOCKS_PER_CU : 1) void chain_kernel(const ChainParams p)
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

    // Philox draws in FAST math: the proposal width is folded into the Box-Muller radius
    constexpr bool PRESCALED = !INJ && sizeof(Real) == 4 && MATH == MCMC_MATH_FAST;
    const Target<Real> tg = make_target<Real>(p.target);
    const Real step = (Real)p.step;
    const float scale2 = -1.3862943611198906f * (float)p.step * (float)p.step;
    (void)scale2;
    const Real center = (Real)p.center;
    const uint64_t t_begin = p.step_offset;
    const uint64_t t_end = p.step_offset + p.n_steps;
    const uint64_t nc = p.n_chains;

    // the per-lane FP64 sums of y, y^2 run across the block's tiles and are reduced once at the end
    double S1acc = 0.0, S2acc = 0.0;
    uint64_t wacc_all = 0;

    uint64_t stamp_c = 0, stamp_r = 0;
    if constexpr (STAMP) {
        stamp_c = __builtin_amdgcn_s_memtime();
        stamp_r = __builtin_amdgcn_s_memrealtime();
    }

    for (uint32_t tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const uint64_t chain = (uint64_t)tile * CHAIN_BLOCK + tid;
        const bool valid = chain < nc;
        // A wave that lies wholly past the end (last tile only) has nothing to do.  It must not replay a chain:
        // waves of a block are not synchronised inside this loop, so it could load that chain's state after the
        // owning wave has stored the advanced one and write a different trajectory into the owner's slots.
        if ((uint64_t)tile * CHAIN_BLOCK + (uint64_t)wave * 64 >= nc) continue;
        // Lanes past the end in the SAME wave as the last chain replay it: same start (one load instruction, ahead
        // of the wave's own store), same draws, same trajectory -- so their per-step stores hit that chain's slots
        // with that chain's values and need no exec masking; only the statistics leave them out.
        const uint64_t ch = valid ? chain : nc - 1;
        const uint64_t gid = p.chain_offset + ch;
        Real *state = (Real *)p.state + (uint64_t)coord * nc + ch;
        const uint32_t lane_off_b = (uint32_t)ch * (uint32_t)sizeof(Real); // STORE == 2: n_chains < 2^30 (launch_chain)
        const uint32_t row_bytes = (uint32_t)nc * (uint32_t)sizeof(Real);  // one row samples[t][coord][0 .. nc)
        (void)lane_off_b, (void)row_bytes;
        float x_even = 0.0f; // STORE == 3: the sample of the pair's first step
        (void)x_even;

        // this chain's draw stream: the block-independent part of Philox rounds 1-2 (device_common.hpp).  Not with the
        // lag products: that kernel sits at its register cap and the five words more cost it 10 % (measured).
        constexpr bool STREAM = NLAG == 0 && !INJ;
        PhiloxStream pstream = {};
        if constexpr (STREAM) pstream = philox_stream(p.seed, gid, coord, philox_cw(3 * (t_begin >> 3), TAG_MH));
        (void)pstream;

        Real x = *state;
        Real aux;          // float: 1/p | p | log p | d or g (CROSS); double: p(x)
        float aux2 = 0.0f; // CROSS: n
        if constexpr (sizeof(Real) == 4) mh_aux_init<KIND, RULE, MATH>(tg, x, aux, aux2);
        else aux = pdf_precise<KIND>(tg, x);

        // Lag history: between steps H[LAGPAD + k] = y_{t-1-k}.  The LAGPAD slots in front let a group of
        // LAGPAD steps write its samples at static indices and shift the window once per group
        // (NL/LAGPAD register moves per step instead of NL).  Kept in float: it only feeds the float
        // lag products and the head/tail sums.
        // Both arrays are register PAIRS (HN, NP even): on the even steps of a group the history pair
        // (H[B+k], H[B+k+1]) = (lag k+1, lag k+2) is an aligned pair like the accumulator pair (P_k, P_k+1), so two
        // products are ONE v_pk_fma_f32 (4.2 issue cycles against 2 x 3.1 for v_fmac_f32, tools/ubench_cal.hip); on odd
        // steps the pairs straddle and the products are single FMAs.  Every P_k receives the same FMAs in the same
        // order either way.  (A circular window with static slots -- no shifts at all -- needs the loop unrolled over
        // 56 steps: 36 KB of code, and measured SLOWER, 312 against 300 cycles per proposal.)
        constexpr int NP = (NL + 1) / 2;              // accumulator pairs; the odd tail lane is a dummy
        constexpr int HN = (NL + LAGPAD + 2) / 2;     // history pairs (one spare slot feeds the dummy lane)
        f32x2 H2[HN];
        f32x2 P2[NP];
#define MCMC_LAGH(j) H2[(j) >> 1][(j)&1]
#define MCMC_LAGP(k) P2[(k) >> 1][(k)&1]
        // FP64 instantiation (metropolis.hpp / statistics.hpp arithmetic): the float lag products are folded into
        // FP64 at every flush (<= 128 steps), like s1 / s2.  The float kernels keep P in FP32 for the launch
        // (49 more FP64 accumulators would halve their occupancy): see mcmc_hip.h for the bound this puts on n_steps.
        constexpr int NPD = (NLAG > 0 && sizeof(Real) == 8) ? NL : 1;
        double PD[NPD];
#pragma unroll
        for (int k = 0; k < NPD; ++k) PD[k] = 0.0;
#pragma unroll
        for (int k = 0; k < NP; ++k) P2[k] = f32x2{0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < HN; ++k) H2[k] = f32x2{0.0f, 0.0f};
        if (NLAG > 0 && !p.window_start && p.ring != nullptr) {
#pragma unroll
            for (int k = 0; k < NL; ++k)
                if (k < (int)p.n_lags)
                    MCMC_LAGH(LAGPAD + k) = (float)((const Real *)p.ring)[((uint64_t)coord * p.n_lags + k) * nc + ch];
        }

        double S1 = 0.0, S2 = 0.0;
        float s1 = 0.0f, s2 = 0.0f;
        // accepted proposals are counted per wave on the scalar unit (ballot + s_bcnt1): no VALU slot
        const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid);
        uint64_t wacc = 0;
        uint64_t cursor = 0; // injected uniforms consumed (rule SHORTCIRCUIT)
        (void)cursor;

        // One sweep of this chain: global step t.
        // SLOT < 0: generic step (shifts the lag window by one).  SLOT = s in [0, LAGPAD): step s of a
        // run of LAGPAD steps, window shifted once by the caller after the run.
        auto one_step = [&](auto slot_tag, uint64_t t, Real normal, Real uniform) __attribute__((always_inline)) {
            constexpr int SLOT = decltype(slot_tag)::value;
            uint64_t slot = 0;
            if constexpr (INJ || STORE == 1) slot = ((t - t_begin) * p.n_coords + coord) * nc + ch;
            if constexpr (INJ) {
                normal = ((const Real *)p.inj_normals)[slot];
                if constexpr (RULE != MCMC_ACCEPT_SHORTCIRCUIT) uniform = ((const Real *)p.inj_uniforms)[slot];
            }
            uint64_t took; // the wave's mask of accepting lanes
            if constexpr (PRESCALED) {
                took = mh_step_dz<KIND, RULE, MATH>(tg, x, aux, aux2, normal, uniform); // `normal` is step * z
            } else if constexpr (sizeof(Real) == 4) {
                // Metropolis.cuh:43: mul + add, two roundings
                took = mh_step_dz<KIND, RULE, MATH>(tg, x, aux, aux2, step * normal, uniform);
            } else {
                // metropolis.hpp:36-44
                const Real proposed = x + step * normal;
                const Real pp = pdf_precise<KIND>(tg, proposed);
                bool take = pp > aux;
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
                took = __builtin_amdgcn_ballot_w64(take);
            }
            wacc += (uint64_t)__builtin_popcountll(took & vmask);
            const Real y = x - center;
            s1 += (float)y;
            s2 = __builtin_fmaf((float)y, (float)y, s2);
            if constexpr (NLAG > 0) {
                constexpr int BASE = SLOT < 0 ? LAGPAD : LAGPAD - SLOT;
                if constexpr (SLOT >= 0 && (BASE & 1) == 0 && sizeof(Real) == 4) {
                    const f32x2 yy = {(float)y, (float)y}; // only the low half is read (op_sel / op_sel_hi of src0 = 0)
#pragma unroll
                    for (int jj = 0; jj < NP; ++jj)
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(P2[jj]) : "v"(yy), "v"(H2[(BASE >> 1) + jj]));
                } else {
#pragma unroll
                    for (int k = 0; k < NL; ++k) MCMC_LAGP(k) = __builtin_fmaf((float)y, MCMC_LAGH(BASE + k), MCMC_LAGP(k));
                }
                if constexpr (SLOT < 0) {
#pragma unroll
                    for (int k = NL - 1; k > 0; --k) MCMC_LAGH(LAGPAD + k) = MCMC_LAGH(LAGPAD + k - 1);
                    MCMC_LAGH(LAGPAD) = (float)y;
                } else {
                    MCMC_LAGH(BASE - 1) = (float)y;
                }
            }
            if constexpr (STORE == 2) {
                const Real *rowp = (const Real *)p.samples + ((t - t_begin) * p.n_coords + coord) * nc; // uniform
                store_row_f32(reinterpret_cast<const float *>(rowp), row_bytes, lane_off_b, (float)x);
            }
            if constexpr (STORE == 3) {
                if ((t & 1) == 0) { // t_begin is even: the parity of t is the parity within the launch
                    x_even = (float)x;
                } else {
                    const Real *rowp = (const Real *)p.samples + (((t - t_begin) >> 1) * p.n_coords + coord) * nc * 2; // uniform
                    store_row_f32x2(reinterpret_cast<const float *>(rowp), 2 * row_bytes, 2 * lane_off_b, x_even, (float)x);
                }
            }
            if constexpr (STORE == 1) {
                // written once, never re-read by this kernel: streaming (nt) stores keep them out of L2's way
                if (p.samples != nullptr) __builtin_nontemporal_store(x, (Real *)p.samples + slot);
                if (p.accept_bits != nullptr)
                    __builtin_nontemporal_store((uint8_t)((took >> lane) & 1u), p.accept_bits + slot);
            }
        };

        auto flush = [&]() __attribute__((always_inline)) {
            S1 += (double)s1;
            S2 += (double)s2;
            s1 = 0.0f;
            s2 = 0.0f;
            if constexpr (NLAG > 0 && sizeof(Real) == 8) {
#pragma unroll
                for (int k = 0; k < NL; ++k) {
                    PD[k] += (double)MCMC_LAGP(k);
                    MCMC_LAGP(k) = 0.0f;
                }
            }
        };

        // One group of 8 steps = 4 step pairs from 3 Philox blocks (device_common.hpp, packed draws).
        // FULL: every step of the group is inside [ta, tb): no range tests, static lag-window slots
        // (the window is shifted by LAGPAD = 8, whole register pairs, after the group).  Otherwise (first / last
        // group of a launch) each step is range-tested and shifts the window itself.
        auto do_group = [&](auto full_tag, uint64_t g, uint64_t ta, uint64_t tb) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tag)::value;
            Spare sp = {};
            auto body = [&](auto j_tag) __attribute__((always_inline)) {
                constexpr int J = decltype(j_tag)::value;
                Real z0 = 0, z1 = 0, u0 = 0, u1 = 0;
                if constexpr (!INJ) {
                    // full groups: the caller has checked that pstream covers the group's blocks
                    const u32x4 w = (FULL && STREAM) ? pair_words_stream<J>(pstream, p.seed, g, sp) : pair_words<J>(p.seed, gid, coord, g, sp);
                    if constexpr (sizeof(Real) == 4) {
                        float a, b, c, d;
                        pair_draws_f32<MATH, PRESCALED>(w, scale2, a, b, c, d);
                        z0 = a, z1 = b, u0 = c, u1 = d;
                    } else {
                        double a, b, c, d;
                        pair_draws_f64(w, a, b, c, d);
                        z0 = a, z1 = b, u0 = c, u1 = d;
                    }
                }
                const uint64_t t0 = (g << 3) + 2 * J;
                if constexpr (FULL) {
                    one_step(std::integral_constant<int, (2 * J) & (LAGPAD - 1)>{}, t0, z0, u0);
                    one_step(std::integral_constant<int, (2 * J + 1) & (LAGPAD - 1)>{}, t0 + 1, z1, u1);
                    if constexpr (NLAG > 0 && 2 * J + 2 == LAGPAD) {
                        // slots LAGPAD .. NL+LAGPAD-1 <- slots 0 .. NL-1, pair by pair (LAGPAD is even)
#pragma unroll
                        for (int j = (NL + LAGPAD - 1) >> 1; j >= LAGPAD / 2; --j) H2[j] = H2[j - LAGPAD / 2];
                    }
                } else {
                    constexpr std::integral_constant<int, -1> GEN{};
                    if (t0 >= ta && t0 < tb) one_step(GEN, t0, z0, u0);
                    if (t0 + 1 >= ta && t0 + 1 < tb) one_step(GEN, t0 + 1, z1, u1);
                }
            };
            expand_sequence(body, std::make_integer_sequence<int, 4>{});
        };

        // Steps [ta, tb)
        auto run_steps = [&](uint64_t ta, uint64_t tb) __attribute__((always_inline)) {
            if (ta >= tb) return;
            uint64_t g = ta >> 3;
            const uint64_t g_end = (tb + 7) >> 3; // exclusive
            if ((ta & 7) != 0 || (g << 3) + 8 > tb) {
                do_group(std::false_type{}, g, ta, tb);
                ++g;
            }
            const uint64_t g_full_end = tb >> 3;
            while (g < g_full_end) {
                uint64_t stop = (g | 15u) + 1; // the float partial sums are folded into FP64 every <= 128 steps
                if (stop > g_full_end) stop = g_full_end;
                // the stream's precomputed rounds hold while the block index stays below the next multiple of 2^32
                const uint32_t cw_a = philox_cw(3 * g, TAG_MH), cw_b = philox_cw(3 * stop - 1, TAG_MH);
                if (!STREAM || cw_a == cw_b) {
                    if (STREAM && cw_a != pstream.cw) pstream = philox_stream(p.seed, gid, coord, cw_a);
                    for (; g < stop; ++g) do_group(std::true_type{}, g, ta, tb);
                } else {
                    for (; g < stop; ++g) do_group(std::false_type{}, g, ta, tb); // range-tested steps, generic draws
                }
                flush();
            }
            if (g < g_end) do_group(std::false_type{}, g, ta, tb);
            flush();
        };

        // Phase 0: the first n_lags steps of a window, after which the lag window holds exactly the
        // window's first n_lags samples (the "head" sums A_i).  Phase 1: the rest.  One loop so that
        // the step code exists once.
        uint64_t t_mid = t_begin;
        if constexpr (NLAG > 0) {
            if (p.window_start && p.n_steps >= (uint64_t)p.n_lags) t_mid = t_begin + p.n_lags;
        }
        if constexpr (NLAG == 0) {
            // no phases: inside the phase loop the draws of the launch's first (range-tested) group are loop-invariant,
            // and the compiler computed them at the top of every tile whether that group ran or not (224 VALU
            // instructions per tile: 2 % of a 256-sweep launch)
            run_steps(t_begin, t_end);
        } else {
#pragma unroll 1
            for (int phase = 0; phase < 2; ++phase) {
                run_steps(phase == 0 ? t_begin : t_mid, phase == 0 ? t_mid : t_end);
                if (phase == 0 && t_mid != t_begin) {
#pragma unroll
                    for (int k = 0; k < NL; ++k) {
                        if (k < (int)p.n_lags) {
                            // H[LAGPAD + k] = y_{n_lags-1-k}
                            const double w = wave_sum(valid ? (double)MCMC_LAGH(LAGPAD + k) : 0.0);
                            if (lane == 0) sred[wave][3 + p.n_lags + (p.n_lags - 1 - k)] += (double)w;
                        }
                    }
                }
            }
        }

        if (valid) *state = x;

        if (NLAG > 0 && p.ring != nullptr && valid) {
#pragma unroll
            for (int k = 0; k < NL; ++k)
                if (k < (int)p.n_lags)
                    ((Real *)p.ring)[((uint64_t)coord * p.n_lags + k) * nc + ch] = (Real)MCMC_LAGH(LAGPAD + k);
        }

        S1acc += valid ? S1 : 0.0;
        S2acc += valid ? S2 : 0.0;
        wacc_all += wacc;
        // lag sums: wave reductions -> per-wave LDS accumulators (only lane 0 of a wave touches its row)
        if constexpr (NLAG > 0) {
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                if (k < (int)p.n_lags) {
                    // FP64 across chains: sums must not depend on how chains fall into waves / GPUs
                    double pk_lane = (double)MCMC_LAGP(k);
                    if constexpr (sizeof(Real) == 8) pk_lane += PD[k];
                    const double pk = wave_sum(valid ? pk_lane : 0.0);
                    const double zk = wave_sum(valid ? (double)MCMC_LAGH(LAGPAD + k) : 0.0);
                    if (lane == 0) {
                        sred[wave][3 + k] += (double)pk;
                        sred[wave][3 + 2 * p.n_lags + k] += (double)zk;
                    }
                }
            }
        }
    }

    {
        const double r1 = wave_sum(S1acc);
        const double r2 = wave_sum(S2acc);
        if (lane == 0) {
            sred[wave][0] = r1;
            sred[wave][1] = r2;
            sred[wave][2] = (double)wacc_all;
        }
    }
    __syncthreads();
    if (tid < len) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < CHAIN_WAVES; ++w) tot += sred[w][tid];
        p.partials[((uint64_t)coord * gridDim.x + blockIdx.x) * len + tid] = tot;
    }
#undef MCMC_LAGP
#undef MCMC_LAGH
    if constexpr (STAMP) {
        const uint64_t dc = __builtin_amdgcn_s_memtime() - stamp_c;
        const uint64_t dr = __builtin_amdgcn_s_memrealtime() - stamp_r;
        if (tid == 0 && p.stamps != nullptr) {
            uint64_t *o = p.stamps + ((uint64_t)coord * gridDim.x + blockIdx.x) * 2;
            o[0] = dc;
            o[1] = dr;
        }
    }
}

// Fixed-order sum of the block partials: grid = (len, n_coords); thread t adds partials
// t, t+256, ... in order, then a fixed LDS tree -- the result does not depend on timing.
static __global__ __launch_bounds__(256) void chain_finalize_kernel(const double *partials, double *stats, int len,
                                                                  int nblocks)
{
    __shared__ double sh[256];
    const int i = blockIdx.x;
    const int coord = blockIdx.y;
    double tot = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) tot += partials[((uint64_t)coord * nblocks + b) * len + i];
    sh[threadIdx.x] = tot;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) stats[(uint64_t)coord * len + i] = sh[0];
}

// the stamped twin of the FAST-math Philox kernels without write-back (what bench.py times)
template <int KIND, int RULE, int NLAG>
void launch_chain_stamped(const ChainParams &p, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL((chain_kernel<float, KIND, RULE, MCMC_MATH_FAST, NLAG, false, 0, true>), grid, dim3(CHAIN_BLOCK), 0,
                       s, p);
}

template <typename Real, int KIND, int RULE, int MATH, int NLAG, bool INJ>
void launch_chain(const ChainParams &p, dim3 grid, hipStream_t s)
{
    if constexpr (sizeof(Real) == 4 && MATH == MCMC_MATH_FAST && NLAG == 0 && !INJ) {
        if (p.samples != nullptr && p.samples_pairs) { // chain_launch has checked the layout's conditions
            hipLaunchKernelGGL((chain_kernel<Real, KIND, RULE, MATH, NLAG, INJ, 3>), grid, dim3(CHAIN_BLOCK), 0, s, p);
            return;
        }
        if (p.samples != nullptr && p.accept_bits == nullptr && p.n_chains < (1ull << 30)) {
            hipLaunchKernelGGL((chain_kernel<Real, KIND, RULE, MATH, NLAG, INJ, 2>), grid, dim3(CHAIN_BLOCK), 0, s, p);
            return;
        }
    }
    if (p.samples != nullptr || p.accept_bits != nullptr)
        hipLaunchKernelGGL((chain_kernel<Real, KIND, RULE, MATH, NLAG, INJ, 1>), grid, dim3(CHAIN_BLOCK), 0, s, p);
    else
        hipLaunchKernelGGL((chain_kernel<Real, KIND, RULE, MATH, NLAG, INJ, 0>), grid, dim3(CHAIN_BLOCK), 0, s, p);
}

} // namespace mcmc
