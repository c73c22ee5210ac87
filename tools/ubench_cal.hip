// tools/ubench_cal.hip -- calibration loops for the VALU counters (profiles/r2_valu_busy_calibrated.json).
// Each kernel is a pure stream of ONE instruction (inline asm, 8 independent chains per lane, 8 waves per SIMD on every
// CU), so its VALU issue occupancy is 100 % by construction; run under rocprofv3 --pmc it shows what the SQ counters
// read at 100 % and in which SQ_INSTS_VALU_* class the instruction is tallied, and its kernel time gives the issue cost
// in SIMD cycles per wave64 instruction (clock from GRBM_GUI_ACTIVE of the same dispatch).  One kernel name per
// instruction, so one profiler run covers them all.  Build: make ubench.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess) {                                                   \
            std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
            std::exit(1);                                                        \
        }                                                                        \
    } while (0)

constexpr int ITERS = 16384;
constexpr int ILP = 8;
constexpr int UNROLL = 4;

struct op_fma { static constexpr const char *name = "v_fma_f32"; };
struct op_fmac { static constexpr const char *name = "v_fmac_f32"; };
struct op_add { static constexpr const char *name = "v_add_f32"; };
struct op_mul { static constexpr const char *name = "v_mul_f32"; };
struct op_exp { static constexpr const char *name = "v_exp_f32"; };
struct op_log { static constexpr const char *name = "v_log_f32"; };
struct op_rcp { static constexpr const char *name = "v_rcp_f32"; };
struct op_mad64 { static constexpr const char *name = "v_mad_u64_u32"; };
struct op_bitop3 { static constexpr const char *name = "v_bitop3_b32"; };
struct op_xor { static constexpr const char *name = "v_xor_b32"; };
struct op_alignbit { static constexpr const char *name = "v_alignbit_b32"; };
struct op_perm { static constexpr const char *name = "v_perm_b32"; };
struct op_cmp { static constexpr const char *name = "v_cmp_ge_f32"; };
struct op_cndmask { static constexpr const char *name = "v_cndmask_b32"; };
struct op_mov { static constexpr const char *name = "v_mov_b32"; };
struct op_max { static constexpr const char *name = "v_max_f32"; };
struct op_mulabs { static constexpr const char *name = "v_mul_f32 |src|"; };
// the accept / select pattern of the MH step and its alternatives (per op: the listed instructions)
struct op_cnd_sgpr { static constexpr const char *name = "v_cndmask_b32_e64 (mask in an SGPR pair)"; static constexpr int n = 1; };
struct op_cnd_vccset { static constexpr const char *name = "v_cndmask_b32_e32 (vcc written once before the loop)"; static constexpr int n = 1; };
struct op_cmp_cnd3 { static constexpr const char *name = "v_cmp + 3 v_cndmask"; static constexpr int n = 4; };
struct op_cmp_fma_cnd3 { static constexpr const char *name = "v_cmp, 4 v_fma, 3 v_cndmask, 4 v_fma"; static constexpr int n = 12; };
struct op_cmpx_mov3 { static constexpr const char *name = "v_cmpx (exec) + 3 v_mov + s_mov exec"; static constexpr int n = 4; };
struct op_sub_ashr_bitop3 { static constexpr const char *name = "v_sub, v_ashrrev, 3 v_bitop3 0xCA (sign-mask select)"; static constexpr int n = 5; };
struct op_cmp_sgpr_cnd3 { static constexpr const char *name = "v_cmp_e64 -> SGPR pair + 3 v_cndmask_e64"; static constexpr int n = 4; };
struct op_pk_fma { static constexpr const char *name = "v_pk_fma_f32 (2 FMAs)"; static constexpr int n = 1; };
struct op_pk_fma_bcast { static constexpr const char *name = "v_pk_fma_f32 op_sel broadcast of src0.lo (2 FMAs)"; static constexpr int n = 1; };
struct op_dot2_f16 { static constexpr const char *name = "v_dot2_f32_f16 (2 MACs, f32 accumulate)"; static constexpr int n = 1; };
struct op_dot2c_f16 { static constexpr const char *name = "v_dot2c_f32_f16 (2 MACs, f32 accumulate, VOP2)"; static constexpr int n = 1; };
struct op_and_or { static constexpr const char *name = "v_and_or_b32"; static constexpr int n = 1; };
struct op_lshr { static constexpr const char *name = "v_lshrrev_b32"; static constexpr int n = 1; };
struct op_cvt_u32 { static constexpr const char *name = "v_cvt_f32_u32"; static constexpr int n = 1; };
struct op_bfe { static constexpr const char *name = "v_bfe_u32"; static constexpr int n = 1; };
struct op_fma12 { static constexpr const char *name = "12 v_fma (reference for the mixed loop)"; static constexpr int n = 12; };

template <typename OP>
__global__ __launch_bounds__(256) void cal(unsigned *out, unsigned seed)
{
    unsigned a[ILP], b[ILP];
    unsigned long long w[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
        a[i] = seed + threadIdx.x * 2654435761u + i * 7919u;
        b[i] = a[i] * 31u + 1u;
        w[i] = a[i];
    }
    const unsigned c = seed * 3u + 0x3f800000u, m = 0xD2511F53u;
    unsigned lo32[ILP], hi32[ILP], f0[ILP], f1[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) lo32[i] = threadIdx.x + i, hi32[i] = threadIdx.x * 3u + i, f0[i] = 0x3f000000u + i, f1[i] = 0x3e800000u + i;
    unsigned long long mask = 0x5555aaaa0f0f3333ull ^ seed;
    unsigned long long w2[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) w2[i] = 0x3f0000003e800000ull + i;
    const unsigned long long w3 = 0x3f8000003f800001ull + seed;
    if constexpr (__is_same(OP, op_cnd_vccset)) asm volatile("v_cmp_ge_f32_e32 vcc, %0, %1" : : "v"(a[0]), "v"(b[0]) : "vcc");
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < UNROLL; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if constexpr (__is_same(OP, op_fma)) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c));
                if constexpr (__is_same(OP, op_fmac)) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));
                if constexpr (__is_same(OP, op_add)) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if constexpr (__is_same(OP, op_mul)) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if constexpr (__is_same(OP, op_mulabs)) asm volatile("v_mul_f32_e64 %0, %1, |%0|" : "+v"(a[i]) : "v"(b[i]));
                if constexpr (__is_same(OP, op_exp)) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(a[i]));
                if constexpr (__is_same(OP, op_log)) asm volatile("v_log_f32_e32 %0, %0" : "+v"(a[i]));
                if constexpr (__is_same(OP, op_rcp)) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[i]));
                if constexpr (__is_same(OP, op_mad64)) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "s"(m) : "vcc");
                if constexpr (__is_same(OP, op_bitop3)) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b[i]), "v"(c));
                if constexpr (__is_same(OP, op_xor)) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if constexpr (__is_same(OP, op_alignbit)) asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a[i]) : "v"(c));
                if constexpr (__is_same(OP, op_perm)) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b[i]), "s"(0x04000201u));
                if constexpr (__is_same(OP, op_cmp)) asm volatile("v_cmp_ge_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
                if constexpr (__is_same(OP, op_cndmask)) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b[i]) :);
                if constexpr (__is_same(OP, op_mov)) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                if constexpr (__is_same(OP, op_max)) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if constexpr (__is_same(OP, op_pk_fma)) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(w[i]) : "v"(w2[i]), "v"(w3));
                if constexpr (__is_same(OP, op_pk_fma_bcast))
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(w[i]) : "v"(w3), "v"(w2[i]));
                if constexpr (__is_same(OP, op_dot2_f16)) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c));
                if constexpr (__is_same(OP, op_dot2c_f16)) asm volatile("v_dot2c_f32_f16_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));
                if constexpr (__is_same(OP, op_and_or)) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));
                if constexpr (__is_same(OP, op_lshr)) asm volatile("v_lshrrev_b32_e32 %0, 9, %0" : "+v"(a[i]));
                if constexpr (__is_same(OP, op_cvt_u32)) asm volatile("v_cvt_f32_u32_e32 %0, %0" : "+v"(a[i]));
                if constexpr (__is_same(OP, op_bfe)) asm volatile("v_bfe_u32 %0, %0, 9, 23" : "+v"(a[i]));
                if constexpr (__is_same(OP, op_cnd_sgpr)) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b[i]), "s"(mask));
                if constexpr (__is_same(OP, op_cnd_vccset)) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b[i]) :);
                if constexpr (__is_same(OP, op_cmp_cnd3))
                    asm volatile("v_cmp_ge_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %1, %0, vcc\n\t"
                                 "v_cndmask_b32_e32 %2, %3, %2, vcc\n\tv_cndmask_b32_e32 %3, %1, %3, vcc"
                                 : "+v"(a[i]), "+v"(b[i]), "+v"(lo32[i]), "+v"(hi32[i]) : : "vcc");
                if constexpr (__is_same(OP, op_cmp_fma_cnd3))
                    asm volatile("v_cmp_ge_f32_e32 vcc, %0, %1\n\t"
                                 "v_fma_f32 %4, %4, %6, %4\n\tv_fma_f32 %5, %5, %6, %5\n\tv_fma_f32 %4, %4, %6, %4\n\tv_fma_f32 %5, %5, %6, %5\n\t"
                                 "v_cndmask_b32_e32 %0, %1, %0, vcc\n\tv_cndmask_b32_e32 %2, %3, %2, vcc\n\tv_cndmask_b32_e32 %3, %1, %3, vcc\n\t"
                                 "v_fma_f32 %4, %4, %6, %4\n\tv_fma_f32 %5, %5, %6, %5\n\tv_fma_f32 %4, %4, %6, %4\n\tv_fma_f32 %5, %5, %6, %5"
                                 : "+v"(a[i]), "+v"(b[i]), "+v"(lo32[i]), "+v"(hi32[i]), "+v"(f0[i]), "+v"(f1[i]) : "v"(c) : "vcc");
                if constexpr (__is_same(OP, op_fma12))
                    asm volatile("v_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %1, %1, %2, %1\n\tv_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %1, %1, %2, %1\n\t"
                                 "v_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %1, %1, %2, %1\n\tv_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %1, %1, %2, %1\n\t"
                                 "v_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %1, %1, %2, %1\n\tv_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %1, %1, %2, %1"
                                 : "+v"(f0[i]), "+v"(f1[i]) : "v"(c));
                if constexpr (__is_same(OP, op_cmpx_mov3))
                    asm volatile("v_cmpx_ge_f32_e32 %0, %1\n\tv_mov_b32_e32 %0, %1\n\tv_mov_b32_e32 %2, %3\n\tv_mov_b32_e32 %3, %1\n\t"
                                 "s_mov_b64 exec, -1"
                                 : "+v"(a[i]), "+v"(b[i]), "+v"(lo32[i]), "+v"(hi32[i]) : : "vcc");
                if constexpr (__is_same(OP, op_sub_ashr_bitop3))
                    asm volatile("v_sub_f32_e32 %4, %0, %1\n\tv_ashrrev_i32_e32 %4, 31, %4\n\t"
                                 "v_bitop3_b32 %0, %4, %1, %0 bitop3:0xca\n\tv_bitop3_b32 %2, %4, %3, %2 bitop3:0xca\n\t"
                                 "v_bitop3_b32 %3, %4, %1, %3 bitop3:0xca"
                                 : "+v"(a[i]), "+v"(b[i]), "+v"(lo32[i]), "+v"(hi32[i]), "+v"(f0[i]));
                if constexpr (__is_same(OP, op_cmp_sgpr_cnd3))
                    asm volatile("v_cmp_ge_f32_e64 %4, %0, %1\n\tv_cndmask_b32_e64 %0, %1, %0, %4\n\t"
                                 "v_cndmask_b32_e64 %2, %3, %2, %4\n\tv_cndmask_b32_e64 %3, %1, %3, %4"
                                 : "+v"(a[i]), "+v"(b[i]), "+v"(lo32[i]), "+v"(hi32[i]), "+s"(mask));
            }
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc += a[i] + (unsigned)w[i] + (unsigned)(w[i] >> 32) + lo32[i] + hi32[i] + f0[i] + f1[i];
    acc += (unsigned)mask;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc += (unsigned)w2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename OP>
void run(unsigned *d_out, int blocks)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(cal<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(cal<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double per_simd = 8.0 * ITERS * UNROLL * ILP; // wave-instructions per SIMD and launch
    std::printf("{\"op\": \"%s\", \"ms\": %.4f, \"wave_insts_per_simd\": %.0f, \"ns_per_wave_inst_per_simd\": %.5f}\n", OP::name,
                ms / 3, per_simd, ms / 3 * 1e6 / per_simd);
}

int main()
{
    const int blocks = 256 * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD, one round
    unsigned *d_out;
    CHECK(hipMalloc(&d_out, sizeof(unsigned) * blocks * 256));
    run<op_fma>(d_out, blocks);
    run<op_fmac>(d_out, blocks);
    run<op_add>(d_out, blocks);
    run<op_mul>(d_out, blocks);
    run<op_mulabs>(d_out, blocks);
    run<op_exp>(d_out, blocks);
    run<op_log>(d_out, blocks);
    run<op_rcp>(d_out, blocks);
    run<op_mad64>(d_out, blocks);
    run<op_bitop3>(d_out, blocks);
    run<op_xor>(d_out, blocks);
    run<op_alignbit>(d_out, blocks);
    run<op_perm>(d_out, blocks);
    run<op_cmp>(d_out, blocks);
    run<op_cndmask>(d_out, blocks);
    run<op_mov>(d_out, blocks);
    run<op_max>(d_out, blocks);
    run<op_pk_fma>(d_out, blocks);
    run<op_pk_fma_bcast>(d_out, blocks);
    run<op_dot2_f16>(d_out, blocks);
    run<op_dot2c_f16>(d_out, blocks);
    run<op_and_or>(d_out, blocks);
    run<op_lshr>(d_out, blocks);
    run<op_cvt_u32>(d_out, blocks);
    run<op_bfe>(d_out, blocks);
    run<op_cnd_sgpr>(d_out, blocks);
    run<op_cnd_vccset>(d_out, blocks);
    run<op_cmp_cnd3>(d_out, blocks);
    run<op_cmp_fma_cnd3>(d_out, blocks);
    run<op_fma12>(d_out, blocks);
    run<op_cmpx_mov3>(d_out, blocks);
    run<op_sub_ashr_bitop3>(d_out, blocks);
    run<op_cmp_sgpr_cnd3>(d_out, blocks);
    CHECK(hipFree(d_out));
    return 0;
}
