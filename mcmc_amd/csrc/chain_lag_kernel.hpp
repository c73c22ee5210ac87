// chain_lag_kernel.hpp -- the FP32 many-chain MH kernel WITH lag products (statistics.hpp:36-57, main.cpp:27-30:
// lags 1..49), the lag products on the matrix cores.
//
// The pooled lag product P_k = sum over chains c and steps t of y_t(c) y_{t-k}(c) IS a dense contraction -- over
// the chain index: with Y[t][c] the samples of a wave's 64 chains, G = Y Y^T has G[t][t'] = sum_c y_t(c) y_t'(c) and
// P_k is the sum of G's k-th sub-diagonal.  chain_kernel.hpp forms the 49 products per chain and step on the VALU
// (49 FMAs beside the ~36 instructions of the MH step: 0.53 T proposals/s against 1.30 without lags).  Here:
//   * every step a lane (= chain) splits y into two halves, y s = h + l (h = the top 11 significand bits, l the rest
//     rounded to 11 bits; s a power of two that keeps both in f16's normal range): 22-23 bits of y, three VALU
//     instructions (v_and, v_sub, v_cvt_pk_f16_f32), and writes the pair as one dword to the wave's LDS ring
//     ybuf[t & 31][chain] -- the transposition from "chain per lane" to "time per lane";
//   * every 16 steps the wave reads the block back in MFMA operand layout (lane = (time row r, chain group g): eight
//     f16 = (h, l) of four chains, one ds_read_b128 per 16 chains) once as A (rows = times 16J .. 16J+15) and once as
//     the shifted window S_J (rows = times 16J-1 .. 16J+14), and accumulates, for d = 0..3,
//         C_d[m][n] += sum_c y_{16J+m}(c) y_{16(J-d)-1+n}(c)        (lag = 16 d + 1 + m - n)
//     with v_mfma_f32_16x16x32_f16: A . S gives h h' + l l', rot16(A) . S gives l h' + h l' (K = 32 = 16 chains x 2
//     halves), so C_d receives (h + l)(h' + l') -- the product to 2^-21 relative.  The windows S_{J-1..J-3} stay in
//     registers (48 VGPRs); shifting the windows by one sample makes FOUR tiles cover the lags 1..49 exactly (every
//     lag has 16 entries per block; entries with lag < 1 or > 49 are never read);
//   * the tiles accumulate in f32 over 64 blocks (1024 steps: 2048 MFMA accumulations per entry), then their
//     diagonals are summed through LDS in FP64 (lane k-1 adds the 16 entries of lag k in a fixed order);
//   * the 32 MFMAs of a block are spread over the 16 steps of the NEXT block, two after every step (the block's rows
//     stay in the other half of the ring meanwhile): a wave's matrix instructions then sit between its own vector
//     instructions.  As one burst of 32 per block they cost 194 cycles per wave-step, spread out 186 - 189.
// Per step that is 5 + 1 (ds_write) vector instructions and 2 MFMAs in place of 49 FMAs.  Head / tail sums are MFMA
// row sums (A . 1) of the four blocks at either end of a window; the ring carried between launches and every other
// statistic are those of chain_kernel.hpp; the ring holds exact FP32 samples, so a window split into launches sees the
// same (h, l) pairs as one launch.
#pragma once
#include "chain_kernel.hpp"

namespace mcmc {

typedef _Float16 lag_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 lag_half2 __attribute__((ext_vector_type(2)));
typedef float lag_f32x4 __attribute__((ext_vector_type(4)));

constexpr int LAG_BLK = 16;   // steps per block = M = N of the MFMA
constexpr int LAG_TILES = 4;  // d = 0..3: lags 16 d + 1 + m - n
constexpr int LAG_ROWS = 32;  // LDS ring: two blocks of rows
// Row pitch in dwords: 64 chains + 4.  The operand reads are ds_read_b128 by lane (row r, chain group g) -- a quarter
// wave (fixed g) reads 16 rows at the same column: with a pitch of 64 dwords all 16 fall on the same four banks
// (a 16-way conflict per read, eight reads per block); with 68 they cover all 64 banks.
constexpr int LAG_PITCH = 68;
constexpr int LAG_FLUSH = 64; // blocks between folds of the f32 tiles into FP64
static_assert(16 * (LAG_TILES - 1) + 1 + 15 >= MCMC_MAX_LAGS, "the tiles must reach the largest lag");

// Workgroup: 256 threads, three workgroups per CU (168 VGPRs).  Measured on MI355X (tools/ubench_mfma.hip,
// profiles/r3_ubench_mfma.jsonl): matrix and vector instructions of DIFFERENT waves of a SIMD do not overlap here (a
// burst of 32 MFMAs + 256 v_fma per wave costs their sum), so more waves do not hide the MFMAs.  A 768-thread workgroup
// (the SIMD's three waves entering every block's MFMAs together behind a barrier) measured slower: 246 cycles.
constexpr int LAG_THREADS = CHAIN_BLOCK;
constexpr int LAG_WAVES = LAG_THREADS / 64;
constexpr int LAG_STATS_MAX = 3 + 3 * MCMC_MAX_LAGS;

// workgroups of a launch: one tile of LAG_THREADS chains each, grid-strided beyond 2048 (the partials buffer is sized for <= 2048)
__host__ __device__ inline uint32_t lag_grid_x(uint64_t n_chains)
{
    const uint64_t tiles = (n_chains + LAG_THREADS - 1) / LAG_THREADS;
    return (uint32_t)(tiles < 2048 ? tiles : 2048);
}

// y s -> (h, l) as two f16 in one dword (h low).  The kernel runs with MODE.FP16_OVFL set: |y s| beyond f16's range
// saturates at 65504 instead of becoming inf, so a chain started far from the target cannot poison the tiles with inf * 0.
__device__ __forceinline__ uint32_t lag_pack_scaled(float ys)
{
    const float h = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, ys) & 0xFFFFE000u);
    const float l = ys - h;
    const lag_half2 pk = {(_Float16)h, (_Float16)l}; // v_cvt_pk_f16_f32 (round to nearest even)
    return __builtin_bit_cast(uint32_t, pk);
}

__device__ __forceinline__ uint32_t lag_pack(float y, float scale) { return lag_pack_scaled(y * scale); }

__device__ __forceinline__ void lag_wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int KIND, int RULE, int MATH, bool INJ, int STORE, bool STAMP = false>
__global__ __launch_bounds__(LAG_THREADS, 3) void chain_lag_kernel(const ChainParams p)
{
    constexpr int NL = MCMC_MAX_LAGS;
    __shared__ double sred[LAG_WAVES][LAG_STATS_MAX];
    __shared__ __attribute__((aligned(16))) uint32_t ybuf[LAG_WAVES][LAG_ROWS * LAG_PITCH];

    // MODE.FP16_OVFL (hwreg(HW_REG_MODE, 23, 1)): a conversion to f16 that overflows gives +-65504, not +-inf -- the
    // saturation lag_pack_scaled relies on, without a clamp per step (a v_med3_f32 there measured 2 % of the kernel).
    // The mode belongs to the wave and ends with it.
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1u);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t coord = blockIdx.y;
    const int len = 3 + 3 * (int)p.n_lags;
    const int L = (int)p.n_lags;

    for (int i = tid; i < LAG_WAVES * LAG_STATS_MAX; i += LAG_THREADS) (&sred[0][0])[i] = 0.0;
    __syncthreads();

    constexpr bool PRESCALED = !INJ && MATH == MCMC_MATH_FAST;
    const Target<float> tg = make_target<float>(p.target);
    const float step = (float)p.step;
    const float scale2 = -1.3862943611198906f * (float)p.step * (float)p.step;
    (void)scale2;
    const float center = (float)p.center;
    const uint64_t t_begin = p.step_offset;
    const uint64_t t_end = p.step_offset + p.n_steps;
    const uint64_t nc = p.n_chains;
    // head: the window's first L steps (sums A_i); tail: the launch's last min(L, n_steps) steps (sums Z_j, ring):
    // both are row sums of the blocks that hold them (lag_block), the ring is stored step by step
    const uint64_t head_end = p.window_start ? t_begin + (uint64_t)L : t_begin;
    const uint64_t tail_begin = p.n_steps > (uint64_t)L ? t_end - (uint64_t)L : t_begin;
    const bool carry = !p.window_start && p.ring != nullptr;
    const double inv_s = 1.0 / (double)p.lag_scale;
    const double inv_s2 = inv_s * inv_s;

    uint32_t *yw = &ybuf[wave][0];
    const int r16 = lane & 15, g4 = lane >> 4;

    double S1acc = 0.0, S2acc = 0.0;
    uint64_t wacc_all = 0;

    uint64_t stamp_c = 0, stamp_r = 0;
    if constexpr (STAMP) {
        stamp_c = __builtin_amdgcn_s_memtime();
        stamp_r = __builtin_amdgcn_s_memrealtime();
    }

    const uint32_t n_tiles = (uint32_t)((nc + LAG_THREADS - 1) / LAG_THREADS);
#ifdef MCMC_LAG_PHASES // development build (make lagab): cycles before and after the step loop, summed over the tiles
    uint64_t ph_pro = 0, ph_epi = 0, ph_mark = 0;
#endif
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#ifdef MCMC_LAG_PHASES
        if constexpr (STAMP) ph_mark = __builtin_amdgcn_s_memtime();
#endif
        const uint64_t chain = (uint64_t)tile * LAG_THREADS + tid;
        const bool valid = chain < nc;
        // A wave that lies wholly past the last chain (last tile only) still walks through the loop -- every wave of the
        // workgroup meets the same barriers -- but touches no memory of the chains: it steps a private copy of the last
        // chain and stores nothing (chain_kernel.hpp explains why it must not write that chain's slots).
        const bool wave_live = (uint64_t)tile * LAG_THREADS + (uint64_t)wave * 64 < nc;
        const uint64_t ch = valid ? chain : nc - 1;
        const uint64_t gid = p.chain_offset + ch;
        float *state = (float *)p.state + (uint64_t)coord * nc + ch;
        float *ring = (float *)p.ring + (uint64_t)coord * (uint64_t)L * nc + ch; // ring[k * nc] = y_{t-1-k}
        // lanes past the last chain replay it (chain_kernel.hpp); their samples enter the tiles as zeros
        const float lscale = valid ? p.lag_scale : 0.0f;
        const float ncs = -center * lscale;

        float x = *state;
        float aux, aux2 = 0.0f;
        mh_aux_init<KIND, RULE, MATH>(tg, x, aux, aux2);
        // (no PhiloxStream here: its five words push the kernel over its register cap -- measured in the ISA: 30 spilled
        // registers, some inside the block loop -- and the compiler hoists round 1's fixed product by itself)

        lag_half8 S[LAG_TILES - 1][4]; // shifted windows of the three blocks before the current one
        lag_f32x4 C[LAG_TILES];
#pragma unroll
        for (int d = 0; d < LAG_TILES - 1; ++d)
#pragma unroll
            for (int q = 0; q < 4; ++q) S[d][q] = lag_half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int d = 0; d < LAG_TILES; ++d) C[d] = lag_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        int blocks_since_flush = 0;

#pragma unroll
        for (int row = 0; row < LAG_ROWS; ++row) yw[row * LAG_PITCH + lane] = 0u;

        const uint64_t J0 = t_begin >> 4;
        // history y_tau, tau < t_begin, from the ring of the launch before (0 beyond the ring's L samples)
        auto hist = [&](int64_t tau) __attribute__((always_inline)) -> float {
            const int64_t k = (int64_t)t_begin - 1 - tau;
            return (k >= 0 && k < (int64_t)L) ? ring[(uint64_t)k * nc] : 0.0f;
        };
        if (carry) {
            // the windows S_{J0-3..J0-1}, staged through rows 0..15 of the LDS ring
#pragma unroll 1
            for (int d = LAG_TILES - 1; d >= 1; --d) {
                const int64_t base = (int64_t)16 * ((int64_t)J0 - d) - 1;
#pragma unroll 1
                for (int j = 0; j < LAG_BLK; ++j) yw[j * LAG_PITCH + lane] = lag_pack(hist(base + j), lscale);
                lag_wave_fence();
                lag_half8 w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const lag_half8 *>(yw + r16 * LAG_PITCH + 16 * q + 4 * g4);
                lag_wave_fence();
                // d is a loop variable: select the destination with static indices
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (d == 3) S[2][q] = w[q];
                    else if (d == 2) S[1][q] = w[q];
                    else S[0][q] = w[q];
                }
            }
#pragma unroll
            for (int row = 0; row < LAG_BLK; ++row) yw[row * LAG_PITCH + lane] = 0u;
            // the history inside the first block and the sample before it, at their ring rows
#pragma unroll 1
            for (int j = 0; j < LAG_BLK; ++j) {
                const int64_t tau = (int64_t)16 * (int64_t)J0 - 1 + j;
                if (tau < (int64_t)t_begin) yw[(int)(tau & 31) * LAG_PITCH + lane] = lag_pack(hist(tau), lscale);
            }
            // a launch shorter than the ring: the older entries move up, and are the tail sums beyond its steps
            if (p.n_steps < (uint64_t)L) {
#pragma unroll 1
                for (int j = L - 1; j >= (int)p.n_steps; --j) {
                    const float v = ring[(uint64_t)(j - (int)p.n_steps) * nc];
                    if (valid) ring[(uint64_t)j * nc] = v;
                    const double w = wave_sum(valid ? (double)v : 0.0);
                    if (lane == 0) sred[wave][3 + 2 * L + j] += w;
                }
            }
        }

        double S1 = 0.0, S2 = 0.0;
        float s1 = 0.0f, s2 = 0.0f;
        const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid);
        uint64_t wacc = 0;

        // Fold the f32 tiles into the FP64 lag sums: their diagonals, through LDS.  Staging = rows 0..14 of the ring half
        // `free_half` (0 / 16), whose block has been consumed -- its row 15 may still be the row before a pending block.
        // Two tiles (512 floats) per pass.
        auto flush_tiles = [&](int free_half) __attribute__((always_inline)) {
            float *st = reinterpret_cast<float *>(yw + free_half * LAG_PITCH);
            // The staging addresses depend on the lane alone; left to itself the compiler computes all 32 of them (and 32
            // lane masks) once per kernel, spills them, and reloads them here one by one behind a branch each -- 32
            // dependent trips to scratch memory per fold (measured: 15k - 24k cycles per tile).  An opaque copy of the lane
            // keeps the arithmetic here, and the loop is branch-free.
            int k1 = lane;
            asm volatile("" : "+v"(k1));
            double acc = 0.0;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                lag_wave_fence();
#pragma unroll
                for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[((dd * 16 + 4 * g4 + r) << 4) + r16] = C[2 * pass + dd][r]; // C[i = 4 g + r][j = lane % 16]
                lag_wave_fence();
#pragma unroll
                for (int i = 0; i < LAG_BLK; ++i) {
                    const int v = k1 - i;                // (k - 1) - i with k = lane + 1
                    const int d = (v + 15) >> 4;         // the tile that holds lag k for row i (0 .. 3 for k <= 49)
                    const int j = 16 * d - v;            // its column
                    const bool use = (d >> 1) == pass && k1 < NL;
                    const float val = st[use ? ((((d & 1) * 16 + i) << 4) + j) : 0];
                    acc += use ? (double)val : 0.0;
                }
            }
            if (k1 < L && k1 < NL) sred[wave][3 + k1] += acc * inv_s2;
            lag_wave_fence();
#pragma unroll
            for (int d = 0; d < LAG_TILES; ++d) C[d] = lag_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            blocks_since_flush = 0;
        };

        // Row sums of block J (the sum over the wave's chains of every row, A . 1) -> the head sums A_i and tail sums Z_j
        const _Float16 one_h = (_Float16)1.0f;
        const lag_half8 ones = {one_h, one_h, one_h, one_h, one_h, one_h, one_h, one_h};
        auto add_rows = [&](uint64_t J, const lag_f32x4 &rs) __attribute__((always_inline)) {
            if (r16 == 0) { // column 0: lane 16 g holds the rows 4 g .. 4 g + 3
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint64_t t = (J << 4) + (uint64_t)(4 * g4 + r);
                    const double v = (double)rs[r] * inv_s;
                    if (t >= t_begin && t < head_end) sred[wave][3 + L + (int)(t - t_begin)] += v;
                    if (t >= tail_begin && t < t_end) sred[wave][3 + 2 * L + (int)(t_end - 1 - t)] += v;
                }
            }
        };
        auto wants_rows = [&](uint64_t J) __attribute__((always_inline)) { return (J << 4) < head_end || ((J + 1) << 4) > tail_begin; };

        // Block J = steps [16 J, 16 J + 16) is complete in the LDS ring: its products.  Rows outside [lo, hi) (first / last
        // block of a launch) belong to another launch or do not exist: history / stale as S rows -- where only lags < 1 can
        // reach them -- and zero as A rows.
        auto lag_block = [&](uint64_t J, int lo, int hi) __attribute__((always_inline)) {
            lag_wave_fence();
            const int half = 16 * (int)(J & 1);
            const uint32_t *rowA = yw + (half + r16) * LAG_PITCH + 4 * g4;
            const uint32_t *rowS = yw + ((half + r16 + 31) & 31) * LAG_PITCH + 4 * g4;
            lag_half8 a[4], s0[4];
            // Rows at or after the launch's end hold whatever the ring held before (zeros, or the f32 staging of an earlier
            // fold: any bit pattern, NaN included) and are cleared BEFORE the window is read; rows before its start hold the
            // history, which the window needs, and are cleared after it is read, so that they are zeros as A rows.
            if (hi < LAG_BLK) {
#pragma unroll 1
                for (int row = hi; row < LAG_BLK; ++row) yw[(half + row) * LAG_PITCH + lane] = 0u;
                lag_wave_fence();
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) s0[q] = *reinterpret_cast<const lag_half8 *>(rowS + 16 * q);
            if (lo > 0) {
                lag_wave_fence();
#pragma unroll 1
                for (int row = 0; row < lo; ++row) yw[(half + row) * LAG_PITCH + lane] = 0u;
                lag_wave_fence();
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const lag_half8 *>(rowA + 16 * q);
            if (wants_rows(J)) { // blocks that hold head or tail steps: A . 1, the sum over the wave's chains of every row
                lag_f32x4 rs = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int q = 0; q < 4; ++q) rs = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q], ones, rs, 0, 0, 0);
                add_rows(J, rs);
            }
            // A . S gives h h' + l l', rot16(A) . S gives l h' + h l'.  Four MFMAs in a row on one accumulator (the
            // accumulate-forwarding path: 10.8 instead of 12.3 cycles per MFMA, tools/ubench_mfma.hip).
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int q = 0; q < 4; ++q) C[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q], s0[q], C[0], 0, 0, 0);
#pragma unroll
                for (int d = 1; d < LAG_TILES; ++d)
#pragma unroll
                    for (int q = 0; q < 4; ++q) C[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q], S[d - 1][q], C[d], 0, 0, 0);
                if (pass == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint4 u = __builtin_bit_cast(uint4, a[q]);
                        u.x = __builtin_amdgcn_alignbit(u.x, u.x, 16); // (h, l) -> (l, h)
                        u.y = __builtin_amdgcn_alignbit(u.y, u.y, 16);
                        u.z = __builtin_amdgcn_alignbit(u.z, u.z, 16);
                        u.w = __builtin_amdgcn_alignbit(u.w, u.w, 16);
                        a[q] = __builtin_bit_cast(lag_half8, u);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { // the windows move on by one block
                S[2][q] = S[1][q];
                S[1][q] = S[0][q];
                S[0][q] = s0[q];
            }
            if (++blocks_since_flush >= LAG_FLUSH) flush_tiles(16 - half);
        };

        // One sweep of this chain, global step t; yrow = the LDS ring row of t (+ lane).
        // s1, s2 accumulate y s and (y s)^2 (s a power of two: the same bits as y and y^2, scaled), unscaled at the end
        auto one_step = [&](uint64_t t, uint32_t *yrow, float normal, float uniform) __attribute__((always_inline)) {
            uint64_t slot = 0;
            if constexpr (INJ || STORE == 1) slot = ((t - t_begin) * p.n_coords + coord) * nc + ch;
            if constexpr (INJ) {
                normal = ((const float *)p.inj_normals)[slot];
                uniform = ((const float *)p.inj_uniforms)[slot];
            }
            uint64_t took;
            if constexpr (PRESCALED) took = mh_step_dz<KIND, RULE, MATH>(tg, x, aux, aux2, normal, uniform);
            else took = mh_step_dz<KIND, RULE, MATH>(tg, x, aux, aux2, step * normal, uniform); // Metropolis.cuh:43
            wacc += (uint64_t)__builtin_popcountll(took & vmask);
            const float ys = __builtin_fmaf(x, lscale, ncs); // (x - center) s: one rounding, the scaling is exact
            s1 += ys;
            s2 = __builtin_fmaf(ys, ys, s2);
            *yrow = lag_pack_scaled(ys);
            if constexpr (STORE == 1) {
                if (wave_live) {
                    if (p.samples != nullptr) __builtin_nontemporal_store(x, (float *)p.samples + slot);
                    if (p.accept_bits != nullptr) __builtin_nontemporal_store((uint8_t)((took >> lane) & 1u), p.accept_bits + slot);
                }
            }
        };

        auto flush = [&]() __attribute__((always_inline)) {
            S1 += (double)s1;
            S2 += (double)s2;
            s1 = 0.0f;
            s2 = 0.0f;
        };

        // One group of 8 steps = 4 step pairs from 3 Philox blocks.  FAST: all 8 steps inside the launch (and before the
        // tail when the launch hands a ring on): no range tests.  Otherwise every step is range-tested and, in the tail,
        // stores its sample to the ring.
        // PIPE (fast groups of the pipelined loop): every step is followed by two MFMAs of the block BEFORE the one being
        // stepped through (complete in the other half of the ring): step `slot` of group parity ODD forms tile d = slot & 3
        // of quarter q = 2 ODD + (slot >> 2).  A wave's MFMAs then sit between its own vector instructions instead of
        // forming a burst of 32 during which it issues nothing else.
        lag_half8 pa = {}, pasw = {}, ps0 = {};
        auto do_group = [&](auto fast_tag, auto pipe_tag, auto odd_tag, uint64_t g) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fast_tag)::value;
            constexpr bool PIPE = decltype(pipe_tag)::value;
            constexpr int ODD = decltype(odd_tag)::value;
            Spare sp = {};
            uint32_t *ygrp = yw + (int)((g & 3) << 3) * LAG_PITCH + lane;
            const int halfp = 16 - 16 * (int)((g >> 1) & 1); // the half of the block before the one being stepped
            const uint32_t *prevA = yw + (halfp + r16) * LAG_PITCH + 4 * g4;
            const uint32_t *prevS = yw + ((halfp + r16 + 31) & 31) * LAG_PITCH + 4 * g4;
            auto pipe_slot = [&](auto slot_tag) __attribute__((always_inline)) {
                constexpr int slot = decltype(slot_tag)::value;
                constexpr int q = 2 * ODD + (slot >> 2), d = slot & 3;
                if constexpr (d == 0) {
                    pa = *reinterpret_cast<const lag_half8 *>(prevA + 16 * q);
                    ps0 = *reinterpret_cast<const lag_half8 *>(prevS + 16 * q);
                    uint4 u = __builtin_bit_cast(uint4, pa);
                    u.x = __builtin_amdgcn_alignbit(u.x, u.x, 16);
                    u.y = __builtin_amdgcn_alignbit(u.y, u.y, 16);
                    u.z = __builtin_amdgcn_alignbit(u.z, u.z, 16);
                    u.w = __builtin_amdgcn_alignbit(u.w, u.w, 16);
                    pasw = __builtin_bit_cast(lag_half8, u);
                    C[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, ps0, C[0], 0, 0, 0);
                    C[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pasw, ps0, C[0], 0, 0, 0);
                } else {
                    C[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, S[d - 1][q], C[d], 0, 0, 0);
                    C[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pasw, S[d - 1][q], C[d], 0, 0, 0);
                }
                if constexpr (d == 3) {
                    S[2][q] = S[1][q];
                    S[1][q] = S[0][q];
                    S[0][q] = ps0;
                }
                // nothing but vector / scalar / memory instructions may move across: the MFMAs stay where they are
                __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x20 | 0x40 | 0x80 | 0x100 | 0x200);
            };
            auto body = [&](auto j_tag) __attribute__((always_inline)) {
                constexpr int Jp = decltype(j_tag)::value;
                float z0 = 0, z1 = 0, u0 = 0, u1 = 0;
                if constexpr (!INJ) {
                    const u32x4 w = pair_words<Jp>(p.seed, gid, coord, g, sp);
                    pair_draws_f32<MATH, PRESCALED>(w, scale2, z0, z1, u0, u1);
                }
                const uint64_t t0 = (g << 3) + 2 * Jp;
                if constexpr (FAST) {
                    one_step(t0, ygrp + (2 * Jp) * LAG_PITCH, z0, u0);
                    if constexpr (PIPE) pipe_slot(std::integral_constant<int, 2 * Jp>{});
                    one_step(t0 + 1, ygrp + (2 * Jp + 1) * LAG_PITCH, z1, u1);
                    if constexpr (PIPE) pipe_slot(std::integral_constant<int, 2 * Jp + 1>{});
                } else {
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        const uint64_t t = t0 + o;
                        if (t >= t_begin && t < t_end) {
                            one_step(t, ygrp + (2 * Jp + o) * LAG_PITCH, o ? z1 : z0, o ? u1 : u0);
                            if (p.ring != nullptr && t >= tail_begin && valid) ring[(t_end - 1 - t) * nc] = x - center;
                        }
                    }
                }
            };
            expand_sequence(body, std::make_integer_sequence<int, 4>{});
        };
        constexpr std::true_type YES{};
        constexpr std::false_type NO{};
        constexpr std::integral_constant<int, 0> EVEN{};
        constexpr std::integral_constant<int, 1> ODDG{};

#ifdef MCMC_LAG_PHASES
        if constexpr (STAMP) ph_pro += __builtin_amdgcn_s_memtime() - ph_mark;
#endif
        if (t_begin < t_end) {
            const uint64_t J_last = (t_end - 1) >> 4;
            const uint64_t fast_end_t = p.ring != nullptr ? tail_begin : t_end;
            bool pending = false; // block J - 1 is complete in the ring and its products have not been formed yet
#pragma unroll 1
            for (uint64_t J = J0; J <= J_last; ++J) {
                const bool fast = (J << 4) >= t_begin && ((J + 1) << 4) <= fast_end_t;
                // A block that holds head or tail steps (four at either end of a window) is not pipelined: its row sums.
                // (Pipelining them too -- a third instantiation of the groups, or the row sums read back from the ring --
                // measured slower: the register allocation of the whole loop nest changes, 203 against 194 cycles.)
                const bool rows_prev = pending && wants_rows(J - 1);
                if (fast && pending && !rows_prev) {
                    do_group(YES, YES, EVEN, 2 * J);
                    do_group(YES, YES, ODDG, 2 * J + 1);
                    if (++blocks_since_flush >= LAG_FLUSH) flush_tiles(16 - 16 * (int)(J & 1)); // block J - 1's half (but row 15)
                } else {
                    if (pending) lag_block(J - 1, 0, LAG_BLK);
                    pending = false;
                    if (fast) {
                        do_group(YES, NO, EVEN, 2 * J);
                        do_group(YES, NO, ODDG, 2 * J + 1);
                        pending = true;
                    } else {
                        if ((J << 4) + 8 > t_begin) do_group(NO, NO, EVEN, 2 * J);
                        if ((J << 4) + 8 < t_end) do_group(NO, NO, ODDG, 2 * J + 1);
                        lag_block(J, J == J0 ? (int)(t_begin & 15) : 0, ((J + 1) << 4) <= t_end ? LAG_BLK : (int)(t_end & 15));
                    }
                }
                if ((J & 7) == 7) flush(); // the float partial sums are folded into FP64 every <= 128 steps
            }
            if (pending) lag_block(J_last, 0, LAG_BLK);
#ifdef MCMC_LAG_PHASES
            if constexpr (STAMP) ph_mark = __builtin_amdgcn_s_memtime();
#endif
            flush();
            flush_tiles(0);
        }

        if (valid) *state = x;
        (void)wave_live;

        S1acc += valid ? S1 : 0.0; // in units of s and s^2
        S2acc += valid ? S2 : 0.0;
        wacc_all += wacc;
        lag_wave_fence();
#ifdef MCMC_LAG_PHASES
        if constexpr (STAMP) ph_epi += __builtin_amdgcn_s_memtime() - ph_mark;
#endif
    }

    {
        const double r1 = wave_sum(S1acc) * inv_s;
        const double r2 = wave_sum(S2acc) * inv_s2;
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
        for (int w = 0; w < LAG_WAVES; ++w) tot += sred[w][tid];
        p.partials[((uint64_t)coord * gridDim.x + blockIdx.x) * len + tid] = tot;
    }
    if constexpr (STAMP) {
        const uint64_t dc = __builtin_amdgcn_s_memtime() - stamp_c;
        const uint64_t dr = __builtin_amdgcn_s_memrealtime() - stamp_r;
        if (tid == 0 && p.stamps != nullptr) {
            uint64_t *o = p.stamps + ((uint64_t)coord * gridDim.x + blockIdx.x) * 2;
#ifdef MCMC_LAG_PHASES
            (void)dr;
            o[0] = (ph_pro & 0xffffffffull) | (ph_epi << 32);
            o[1] = dc;
#else
            o[0] = dc;
            o[1] = dr;
#endif
        }
    }
}

} // namespace mcmc
