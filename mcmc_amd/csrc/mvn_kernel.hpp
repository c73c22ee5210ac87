// mvn_kernel.hpp -- the "stress" configuration (BASELINE.json configs[4]):
// many chains on a D-dimensional correlated Gaussian target (D <= 64, multiple of 8) whose
// precision matrix is symmetric tridiagonal -- e.g. Sigma_ij = rho^|i-j| -- with a scalar
// observable written back every sweep, a lag-L autocorrelation accumulator fed from a ring of the
// last L observables in HBM, and optional full-state structure-of-arrays sample write-back.
//
// One sweep is the reference's GibbsStep (xsection/include/Metropolis.cuh:69-79): the coordinates
// are updated one after the other by MetropolisHastingsStep (:40-52), here with the coordinate's
// conditional log-density and LOG-SPACE acceptance: for x_i -> x_i + d
//     log p(x') - log p(x) = -d (h_i + Q_ii d / 2),   h_i = Q_ii x_i + Q_i,i-1 x_i-1 + Q_i,i+1 x_i+1
// and the move is taken when log U <= that difference.
//
// Layout: one chain per lane, the chain's D coordinates in LDS as x[j][lane] (conflict-free, 64 KiB
// per 256-thread block at D = 64, two blocks per CU); a coordinate update reads its right neighbour
// and writes itself.  The band of Q sits in the kernel arguments (scalar loads).  HBM traffic per
// chain and sweep: observable 4 B + ring read 4 B + ring write 4 B (+ 4 D B with full write-back).
// (A variant with the state in registers and the sweep fully unrolled was measured at D = 64:
// 241 VGPRs, same occupancy, 3 % faster without and 2x slower with full write-back -- not kept.)
#pragma once
#include <type_traits>
#include <utility>

#include "device_common.hpp"

namespace mcmc {

constexpr int MVN_BLOCK = 256;
constexpr int MVN_MAX_DIM = 64;
constexpr uint32_t TAG_MVN = 3u;
constexpr int MVN_STATS = 6; // S1, S2, accepts, P_L, HEAD, TAIL

typedef float f8 __attribute__((ext_vector_type(8)));

template <typename F, int... Is>
__device__ __forceinline__ void mvn_expand(F &&f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}

struct MvnParams {
    // band of Q, eight coordinates per vector so that one group of 8 proposals fetches its
    // coefficients with three s_load_dwordx8 (kernel-argument segment) instead of one load per use
    alignas(32) f8 q_diag[MVN_MAX_DIM / 8];
    alignas(32) f8 q_half[MVN_MAX_DIM / 8]; // 0.5 * Q[i][i]
    alignas(32) f8 q_off[MVN_MAX_DIM / 8];  // Q[i][i+1]; 0 for i = dim-1
    uint64_t n_chains, n_steps;
    uint64_t seed, step_offset, chain_offset;
    uint64_t window_begin, window_end;
    uint32_t dim, observable, lag, n_tiles;
    uint32_t ring_row0; // step_offset mod lag
    float step, center;
    float *state;
    float *ring;
    float *samples_obs;
    float *samples_full;
    const float *inj;
    uint8_t *accept_bits;
    double *partials; // [gridDim.x][MVN_STATS]
};

template <int MATH, bool INJ, bool STORE>
__global__ __launch_bounds__(MVN_BLOCK) void mvn_kernel(const MvnParams p)
{
    extern __shared__ __align__(16) float xs[]; // [dim][MVN_BLOCK]
    __shared__ double sred[MVN_BLOCK / 64][MVN_STATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t D = p.dim;
    const uint64_t nc = p.n_chains;
    if (tid < (MVN_BLOCK / 64) * MVN_STATS) (&sred[0][0])[tid] = 0.0;
    __syncthreads();

    const float scale2 = -1.3862943611198906f * p.step * p.step;
    const uint64_t ta = p.step_offset, tb = p.step_offset + p.n_steps;

    for (uint32_t tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const uint64_t chain = (uint64_t)tile * MVN_BLOCK + tid;
        const bool valid = chain < nc;
        const uint64_t ch = valid ? chain : 0;
        const uint64_t gid = p.chain_offset + ch;
        for (uint32_t j = 0; j < D; ++j) xs[j * MVN_BLOCK + tid] = p.state[(uint64_t)j * nc + ch];

        double S1 = 0, S2 = 0, PL = 0, HEAD = 0, TAIL = 0;
        float s1 = 0, s2 = 0, pl = 0, hd = 0, tl = 0;
        const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid);
        uint64_t wacc = 0;

        uint32_t row = p.ring_row0; // t mod lag, carried instead of a 64-bit modulo per sweep
        for (uint64_t t = ta; t < tb; ++t) {
            const uint64_t n0 = t * D; // index of this sweep's first proposal in the chain's draw stream
            // the right neighbour is fetched from LDS one proposal ahead (xp for this proposal, xq for the next):
            // with two waves per SIMD the LDS latency would otherwise sit on the serial chain of the sweep
            float xm = 0.0f, xc = xs[tid], xp = xs[MVN_BLOCK + tid];
            float olm = 0.0f; // Q[i][i-1]
            for (uint32_t gi = 0; gi < D / 8; ++gi) {
                const uint64_t g = (n0 >> 3) + gi;
                const f8 qd8 = p.q_diag[gi], qh8 = p.q_half[gi], qo8 = p.q_off[gi];
                Spare sp = {};
                auto body = [&](auto j_tag) __attribute__((always_inline)) {
                    constexpr int J = decltype(j_tag)::value;
                    float z[2] = {0.0f, 0.0f}, u[2] = {0.0f, 0.0f};
                    if constexpr (!INJ) {
                        const u32x4 w = pair_words_tag<J>(p.seed, gid, 0u, g, TAG_MVN, sp);
                        pair_draws_f32<MATH, MATH == MCMC_MATH_FAST>(w, scale2, z[0], z[1], u[0], u[1]);
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t i = gi * 8 + 2 * J + h;
                        uint64_t slot = 0;
                        if constexpr (INJ || STORE) slot = ((t - ta) * D + i) * nc + ch;
                        float dz;
                        if constexpr (INJ) {
                            const float *d = p.inj + 2 * slot - ch; // [step][coord][2][chain]
                            dz = p.step * d[0];
                            u[h] = d[nc];
                        } else if constexpr (MATH == MCMC_MATH_FAST) {
                            dz = z[h]; // already step * z
                        } else {
                            dz = p.step * z[h];
                        }
                        const float xq = xs[(i + 2 < D ? i + 2 : D - 1) * MVN_BLOCK + tid]; // x_{i+2} (unused at the end: Q = 0)
                        const float qd = qd8[2 * J + h], orr = qo8[2 * J + h];
                        const float hsum = qd * xc + olm * xm + orr * xp;
                        const float t1 = hsum + qh8[2 * J + h] * dz;
                        const float delta = -(dz * t1);
                        float lu;
                        if constexpr (MATH == MCMC_MATH_FAST) lu = 0.6931471805599453f * __builtin_amdgcn_logf(u[h]);
                        else lu = logf(u[h]);
                        // (the EXEC-masked select of device_common.hpp costs this kernel 4 %: two waves per SIMD)
                        const bool take = lu <= delta;
                        const float xn = take ? xc + dz : xc;
                        xs[i * MVN_BLOCK + tid] = xn;
                        wacc += (uint64_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(take) & vmask);
                        if constexpr (STORE) {
                            if (p.accept_bits != nullptr && valid) p.accept_bits[slot] = take ? 1 : 0;
                        }
                        xm = xn;
                        xc = xp;
                        xp = xq;
                        olm = orr;
                    }
                };
                body(std::integral_constant<int, 0>{});
                body(std::integral_constant<int, 1>{});
                body(std::integral_constant<int, 2>{});
                body(std::integral_constant<int, 3>{});
            }
            // the observable of this sweep, its statistics and the lag-L product through the HBM ring
            const float xo = xs[p.observable * MVN_BLOCK + tid];
            const float y = xo - p.center;
            s1 += y;
            s2 = __builtin_fmaf(y, y, s2);
            const uint64_t w = t - p.window_begin;
            float *slot_ring = p.ring + (uint64_t)row * nc + ch;
            row = row + 1 == p.lag ? 0u : row + 1;
            if (w >= p.lag) pl = __builtin_fmaf(y, *slot_ring, pl); // row (t mod L) still holds y_{t-L}
            else hd += y;
            if (t + p.lag >= p.window_end) tl += y;
            if (valid) *slot_ring = y;
            if (p.samples_obs != nullptr && valid) p.samples_obs[(t - ta) * nc + ch] = xo;
            if constexpr (STORE) {
                if (p.samples_full != nullptr && valid)
                    for (uint32_t j = 0; j < D; ++j)
                        p.samples_full[((t - ta) * D + j) * nc + ch] = xs[j * MVN_BLOCK + tid];
            }
            if (((t - ta) & 127u) == 127u) {
                S1 += (double)s1, S2 += (double)s2, PL += (double)pl, HEAD += (double)hd, TAIL += (double)tl;
                s1 = s2 = pl = hd = tl = 0.0f;
            }
        }
        S1 += (double)s1, S2 += (double)s2, PL += (double)pl, HEAD += (double)hd, TAIL += (double)tl;
        if (valid)
            for (uint32_t j = 0; j < D; ++j) p.state[(uint64_t)j * nc + ch] = xs[j * MVN_BLOCK + tid];

        const double r0 = wave_sum(valid ? S1 : 0.0), r1 = wave_sum(valid ? S2 : 0.0), r3 = wave_sum(valid ? PL : 0.0),
                     r4 = wave_sum(valid ? HEAD : 0.0), r5 = wave_sum(valid ? TAIL : 0.0);
        if (lane == 0) {
            sred[wave][0] += r0;
            sred[wave][1] += r1;
            sred[wave][2] += (double)wacc;
            sred[wave][3] += r3;
            sred[wave][4] += r4;
            sred[wave][5] += r5;
        }
    }
    __syncthreads();
    if (tid < MVN_STATS) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < MVN_BLOCK / 64; ++w) tot += sred[w][tid];
        p.partials[(uint64_t)blockIdx.x * MVN_STATS + tid] = tot;
    }
}

// ---------------------------------------------------------------------
// Register-resident twin of mvn_kernel for the stress configuration itself (D = 64, observable = coordinate 0,
// Philox draws, no accept-flag write-back): the chain's 64 coordinates live in VGPRs, the sweep is
// unrolled over them (static register indices), nothing goes through LDS.  LDS capacity (256 B per chain) holds the
// LDS kernel at two waves per SIMD, where its serial dependence x_{i-1} -> x_i is not hidden (rocprofv3: 63-78 % VALU
// issue occupancy, no LDS stall, profiles/r2_valu_busy_calibrated.json); this kernel is compiled for four.
// Same draws, same expressions in the same order: states, statistics and write-back are bit-identical to mvn_kernel's
// (tests/test_gpu_mvn.py compares the two).
constexpr int MVN_REG_DIM = 64;

// FULL: every sweep's 64 coordinates go to samples_full[sweep][coord][chain] (the HBM-heavy variant of the stress
// configuration): streaming stores with a scalar row base and a 32-bit lane offset, issued group by group as the
// coordinates become final, so the sweep's VALU work covers them.
template <int MATH, bool FULL>
__global__ __launch_bounds__(MVN_BLOCK, 4) void mvn_reg_kernel(const MvnParams p)
{
    __shared__ double sred[MVN_BLOCK / 64][MVN_STATS];
    // the band of Q per coordinate as (Q_ii, Q_ii / 2, Q_i,i+1, Q_i,i-1): one broadcast ds_read_b128 per proposal.
    // (Left in the kernel arguments, the unrolled sweep keeps all 192 values in SGPRs and spills them through
    // v_writelane / v_readlane: 5 such instructions per proposal.)
    __shared__ float4 qc[MVN_REG_DIM];
    constexpr int D = MVN_REG_DIM;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint64_t nc = p.n_chains;
    if (tid < (MVN_BLOCK / 64) * MVN_STATS) (&sred[0][0])[tid] = 0.0;
    if (tid < D) {
        const float *qd = reinterpret_cast<const float *>(p.q_diag), *qh = reinterpret_cast<const float *>(p.q_half),
                    *qo = reinterpret_cast<const float *>(p.q_off);
        qc[tid] = make_float4(qd[tid], qh[tid], qo[tid], tid > 0 ? qo[tid - 1] : 0.0f);
    }
    __syncthreads();

    const float scale2 = -1.3862943611198906f * p.step * p.step;
    const uint64_t ta = p.step_offset, tb = p.step_offset + p.n_steps;

    for (uint32_t tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const uint64_t chain = (uint64_t)tile * MVN_BLOCK + tid;
        const bool valid = chain < nc;
        const uint64_t ch = valid ? chain : 0;
        const uint64_t gid = p.chain_offset + ch;
        float xr[D];
        // eight rows at a time.  (With the row addresses shared between these loads and the stores at the end of the
        // tile the kernel spilled ~126 registers to scratch around the sweep loop: 508 B per chain, +1.5 GB of HBM
        // traffic per launch at the stress shape -- rocprofv3 FETCH_SIZE / WRITE_SIZE, profiles/r2_pmc_stress_kernel.json)
#pragma unroll
        for (int j0 = 0; j0 < D; j0 += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) xr[j0 + k] = p.state[(uint64_t)(j0 + k) * nc + ch];
            __builtin_amdgcn_sched_barrier(0);
        }

        double S1 = 0, S2 = 0, PL = 0, HEAD = 0, TAIL = 0;
        float s1 = 0, s2 = 0, pl = 0, hd = 0, tl = 0;
        const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid);
        uint64_t wacc = 0;

        uint32_t row = p.ring_row0;
        // FULL: n_chains < 2^30 (mvn_reg_applies), so a row samples_full[t][i][0 .. nc) and the lane's offset in it fit 32 bits
        // lanes past the last chain get an offset beyond the row: the buffer unit's range check drops their stores
        const uint32_t lane_off_b = valid ? (uint32_t)ch * 4u : 0xFFFFFFFCu, row_bytes = (uint32_t)nc * 4u;
        (void)lane_off_b, (void)row_bytes;
#pragma unroll 1
        for (uint64_t t = ta; t < tb; ++t) {
            const uint64_t n0 = t * D;
            float *full_row = nullptr; // wave-uniform: samples_full[t - ta][0][0]
            if constexpr (FULL) full_row = p.samples_full + (t - ta) * (uint64_t)D * nc;
            (void)full_row;
            auto group = [&](auto gi_tag) __attribute__((always_inline)) {
                constexpr int GI = decltype(gi_tag)::value;
                const uint64_t g = (n0 >> 3) + GI;
                Spare sp = {};
                auto body = [&](auto j_tag) __attribute__((always_inline)) {
                    constexpr int J = decltype(j_tag)::value;
                    float z[2], u[2];
                    const u32x4 w = pair_words_tag<J>(p.seed, gid, 0u, g, TAG_MVN, sp);
                    pair_draws_f32<MATH, MATH == MCMC_MATH_FAST>(w, scale2, z[0], z[1], u[0], u[1]);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        constexpr int I0 = GI * 8 + 2 * J;
                        const int i = I0 + h; // static after unrolling
                        const float dz = MATH == MCMC_MATH_FAST ? z[h] : p.step * z[h];
                        const float xm = i > 0 ? xr[i > 0 ? i - 1 : 0] : 0.0f;
                        const float xc = xr[i];
                        const float xp = xr[i + 1 < D ? i + 1 : D - 1];
                        const float4 q = qc[i]; // Q_ii, Q_ii / 2, Q_i,i+1, Q_i,i-1 (0 for i = 0)
                        const float qd = q.x, orr = q.z, olm = q.w;
                        const float hsum = qd * xc + olm * xm + orr * xp;
                        const float t1 = hsum + q.y * dz;
                        const float delta = -(dz * t1);
                        float lu;
                        if constexpr (MATH == MCMC_MATH_FAST) lu = 0.6931471805599453f * __builtin_amdgcn_logf(u[h]);
                        else lu = logf(u[h]);
                        const bool take = lu <= delta;
                        xr[i] = take ? xc + dz : xc;
                        wacc += (uint64_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(take) & vmask);
                        // coordinate i is final for this sweep (proposal i + 1 only reads it): one unmasked store per
                        // proposal instead of eight under `if (valid)` at the end of a group (0.75 -> 0.79 T proposals/s;
                        // counters of both forms: profiles/r2_store_path_pmc.json)
                        if constexpr (FULL) store_row_f32(full_row + (uint64_t)i * nc, row_bytes, lane_off_b, xr[i]);
                    }
                };
                body(std::integral_constant<int, 0>{});
                body(std::integral_constant<int, 1>{});
                body(std::integral_constant<int, 2>{});
                body(std::integral_constant<int, 3>{});
                // the draws do not depend on the state, so the scheduler would hoist all 24 Philox blocks of a sweep
                // to its top (243 VGPRs): keep each group of 8 proposals to itself
                __builtin_amdgcn_sched_barrier(0);
            };
            mvn_expand(group, std::make_integer_sequence<int, D / 8>{});
            // the observable (coordinate 0), its statistics and the lag-L product through the HBM ring: as mvn_kernel
            const float xo = xr[0];
            const float y = xo - p.center;
            s1 += y;
            s2 = __builtin_fmaf(y, y, s2);
            const uint64_t w = t - p.window_begin;
            float *slot_ring = p.ring + (uint64_t)row * nc + ch;
            row = row + 1 == p.lag ? 0u : row + 1;
            if (w >= p.lag) pl = __builtin_fmaf(y, *slot_ring, pl);
            else hd += y;
            if (t + p.lag >= p.window_end) tl += y;
            if (valid) *slot_ring = y;
            if (p.samples_obs != nullptr && valid) p.samples_obs[(t - ta) * nc + ch] = xo;
            if (((t - ta) & 127u) == 127u) {
                S1 += (double)s1, S2 += (double)s2, PL += (double)pl, HEAD += (double)hd, TAIL += (double)tl;
                s1 = s2 = pl = hd = tl = 0.0f;
            }
        }
        S1 += (double)s1, S2 += (double)s2, PL += (double)pl, HEAD += (double)hd, TAIL += (double)tl;
        if (valid) {
            // the row addresses are formed again here: reusing the 64 of the loads above keeps 128 VGPRs alive across
            // the sweep loop, i.e. in scratch (the empty asm hides `ch` from common-subexpression elimination)
            uint64_t ch_again = ch;
            asm volatile("" : "+v"(ch_again));
#pragma unroll
            for (int j0 = 0; j0 < D; j0 += 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) p.state[(uint64_t)(j0 + k) * nc + ch_again] = xr[j0 + k];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const double r0 = wave_sum(valid ? S1 : 0.0), r1 = wave_sum(valid ? S2 : 0.0), r3 = wave_sum(valid ? PL : 0.0),
                     r4 = wave_sum(valid ? HEAD : 0.0), r5 = wave_sum(valid ? TAIL : 0.0);
        if (lane == 0) {
            sred[wave][0] += r0;
            sred[wave][1] += r1;
            sred[wave][2] += (double)wacc;
            sred[wave][3] += r3;
            sred[wave][4] += r4;
            sred[wave][5] += r5;
        }
    }
    __syncthreads();
    if (tid < MVN_STATS) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < MVN_BLOCK / 64; ++w) tot += sred[w][tid];
        p.partials[(uint64_t)blockIdx.x * MVN_STATS + tid] = tot;
    }
}

} // namespace mcmc
