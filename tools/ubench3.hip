// tools/ubench3.hip -- does the issue cost of a full-rate VALU instruction depend on where its operands live?
// Each loop body is pure inline asm with explicit registers: 8 independent chains (dst v10..v17), 8 waves per SIMD.
//   A: second source in a different VGPR bank (reg % 4) than the destination/source
//   B: same bank      C: second source is an SGPR      D: second source is an inline constant
// cycles anchored on v_exp_f32 = 8.  Build: make ubench.
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

#define CLOBBERS "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "s20", "s21", "s22", "scc", "vcc"
#define INIT                                                                                                         \
    "v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n v_mov_b32 v14, %1\n"            \
    "v_mov_b32 v15, %1\n v_mov_b32 v16, %1\n v_mov_b32 v17, %1\n v_mov_b32 v20, %1\n v_mov_b32 v21, %1\n"            \
    "v_mov_b32 v22, %1\n v_mov_b32 v23, %1\n v_mov_b32 v24, %1\n v_mov_b32 v25, %1\n v_mov_b32 v26, %1\n"            \
    "v_mov_b32 v27, %1\n v_mov_b32 v28, %1\n v_mov_b32 v29, %1\n v_mov_b32 v30, %1\n v_mov_b32 v31, %1\n"            \
    "s_mov_b32 s21, 0x3f800001\n s_mov_b32 s20, %2\n"
#define LOOP_HEAD "1:\n"
#define LOOP_TAIL "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
#define FINI "v_add_u32 %0, v10, v11\n v_add_u32 %0, %0, v12\n v_add_u32 %0, %0, v13\n v_add_u32 %0, %0, v14\n v_add_u32 %0, %0, v15\n v_add_u32 %0, %0, v16\n v_add_u32 %0, %0, v17\n"

// 8 chains, second operand register = dst + OFF  (dst v10..v17)
#define BODY8(OP, S0, S1, S2, S3, S4, S5, S6, S7)                                                                     \
    OP " v10, " S0 ", v10\n " OP " v11, " S1 ", v11\n " OP " v12, " S2 ", v12\n " OP " v13, " S3 ", v13\n "           \
    OP " v14, " S4 ", v14\n " OP " v15, " S5 ", v15\n " OP " v16, " S6 ", v16\n " OP " v17, " S7 ", v17\n"
#define DIFF(OP) BODY8(OP, "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28")  /* dst+11: bank + 3 */
#define SAME(OP) BODY8(OP, "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29")  /* dst+12: same bank */
#define SGPR(OP) BODY8(OP, "s21", "s21", "s21", "s21", "s21", "s21", "s21", "s21")
#define CNST(OP) BODY8(OP, "1.0", "1.0", "1.0", "1.0", "1.0", "1.0", "1.0", "1.0")
#define X4(B) B B B B

#define KERNEL(NAME, BODY)                                                                   \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed)               \
    {                                                                                        \
        unsigned r;                                                                          \
        asm volatile(INIT LOOP_HEAD X4(BODY) LOOP_TAIL FINI : "=v"(r) : "v"(seed + threadIdx.x), "n"(ITERS) : CLOBBERS); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                      \
    }

KERNEL(k_exp, "v_exp_f32 v10, v10\n v_exp_f32 v11, v11\n v_exp_f32 v12, v12\n v_exp_f32 v13, v13\n v_exp_f32 v14, v14\n v_exp_f32 v15, v15\n v_exp_f32 v16, v16\n v_exp_f32 v17, v17\n")
KERNEL(k_xor_diff, DIFF("v_xor_b32"))
KERNEL(k_xor_same, SAME("v_xor_b32"))
KERNEL(k_xor_sgpr, SGPR("v_xor_b32"))
KERNEL(k_xor_cnst, CNST("v_xor_b32"))
KERNEL(k_mul_diff, DIFF("v_mul_f32"))
KERNEL(k_mul_same, SAME("v_mul_f32"))
KERNEL(k_mul_sgpr, SGPR("v_mul_f32"))
KERNEL(k_add_diff, DIFF("v_add_f32"))
KERNEL(k_add_same, SAME("v_add_f32"))
KERNEL(k_addu_diff, DIFF("v_add_u32"))
KERNEL(k_addu_same, SAME("v_add_u32"))
// three-operand forms: fma with all three sources in distinct banks vs two in the same bank
KERNEL(k_fma_3banks, "v_fma_f32 v10, v21, v20, v10\n v_fma_f32 v11, v22, v21, v11\n v_fma_f32 v12, v23, v22, v12\n v_fma_f32 v13, v24, v23, v13\n"
                     "v_fma_f32 v14, v25, v24, v14\n v_fma_f32 v15, v26, v25, v15\n v_fma_f32 v16, v27, v26, v16\n v_fma_f32 v17, v28, v27, v17\n")
KERNEL(k_fma_conflict, "v_fma_f32 v10, v22, v26, v10\n v_fma_f32 v11, v23, v27, v11\n v_fma_f32 v12, v24, v28, v12\n v_fma_f32 v13, v25, v29, v13\n"
                       "v_fma_f32 v14, v26, v30, v14\n v_fma_f32 v15, v27, v31, v15\n v_fma_f32 v16, v28, v20, v16\n v_fma_f32 v17, v29, v21, v17\n")

#define LITR(OP) BODY8(OP, "0x3f800123", "0x3f800123", "0x3f800123", "0x3f800123", "0x3f800123", "0x3f800123", "0x3f800123", "0x3f800123")
KERNEL(k_xor_lit, LITR("v_xor_b32"))
KERNEL(k_sub_lit, LITR("v_sub_f32"))
KERNEL(k_max_diff, DIFF("v_max_f32"))
KERNEL(k_min_diff, DIFF("v_min_f32"))
// 64-bit multiply-add: multiplier in a VGPR / SGPR, destination pairs v[20:21].. (results unused: issue cost only)
KERNEL(k_mad_vgpr, "v_mad_u64_u32 v[20:21], vcc, v10, v30, 0\n v_mad_u64_u32 v[22:23], vcc, v11, v30, 0\n v_mad_u64_u32 v[24:25], vcc, v12, v30, 0\n v_mad_u64_u32 v[26:27], vcc, v13, v30, 0\n"
                   "v_mad_u64_u32 v[20:21], vcc, v14, v30, 0\n v_mad_u64_u32 v[22:23], vcc, v15, v30, 0\n v_mad_u64_u32 v[24:25], vcc, v16, v30, 0\n v_mad_u64_u32 v[26:27], vcc, v17, v30, 0\n")
KERNEL(k_mad_sgpr, "v_mad_u64_u32 v[20:21], vcc, v10, s21, 0\n v_mad_u64_u32 v[22:23], vcc, v11, s21, 0\n v_mad_u64_u32 v[24:25], vcc, v12, s21, 0\n v_mad_u64_u32 v[26:27], vcc, v13, s21, 0\n"
                   "v_mad_u64_u32 v[20:21], vcc, v14, s21, 0\n v_mad_u64_u32 v[22:23], vcc, v15, s21, 0\n v_mad_u64_u32 v[24:25], vcc, v16, s21, 0\n v_mad_u64_u32 v[26:27], vcc, v17, s21, 0\n")
KERNEL(k_mulhi_vgpr, DIFF("v_mul_hi_u32"))
KERNEL(k_mullo_vgpr, DIFF("v_mul_lo_u32"))
#define TRI8(OP, C) OP " v10, v21, v10, " C "\n " OP " v11, v22, v11, " C "\n " OP " v12, v23, v12, " C "\n " OP " v13, v24, v13, " C "\n " \
                    OP " v14, v25, v14, " C "\n " OP " v15, v26, v15, " C "\n " OP " v16, v27, v16, " C "\n " OP " v17, v28, v17, " C "\n"
KERNEL(k_alignbit_const, TRI8("v_alignbit_b32", "9"))
KERNEL(k_perm_vgpr, TRI8("v_perm_b32", "v30"))
KERNEL(k_perm_sgpr, TRI8("v_perm_b32", "s21"))
KERNEL(k_bfi_vgpr, TRI8("v_bfi_b32", "v30"))
KERNEL(k_xad_vgpr, TRI8("v_xad_u32", "v30"))
KERNEL(k_add3_vgpr, TRI8("v_add3_u32", "v30"))
#define TRI8B(C) "v_bitop3_b32 v10, v21, v10, " C " bitop3:0x96\n v_bitop3_b32 v11, v22, v11, " C " bitop3:0x96\n v_bitop3_b32 v12, v23, v12, " C " bitop3:0x96\n v_bitop3_b32 v13, v24, v13, " C " bitop3:0x96\n " \
                "v_bitop3_b32 v14, v25, v14, " C " bitop3:0x96\n v_bitop3_b32 v15, v26, v15, " C " bitop3:0x96\n v_bitop3_b32 v16, v27, v16, " C " bitop3:0x96\n v_bitop3_b32 v17, v28, v17, " C " bitop3:0x96\n"
KERNEL(k_bitop3_vgpr, TRI8B("v30"))
KERNEL(k_bitop3_sgpr, TRI8B("s21"))
KERNEL(k_cmp_vcc, "v_cmp_ge_f32 vcc, v10, v21\n v_cmp_ge_f32 vcc, v11, v22\n v_cmp_ge_f32 vcc, v12, v23\n v_cmp_ge_f32 vcc, v13, v24\n"
                  "v_cmp_ge_f32 vcc, v14, v25\n v_cmp_ge_f32 vcc, v15, v26\n v_cmp_ge_f32 vcc, v16, v27\n v_cmp_ge_f32 vcc, v17, v28\n")
KERNEL(k_cnd_vcc, "v_cndmask_b32 v10, v21, v10, vcc\n v_cndmask_b32 v11, v22, v11, vcc\n v_cndmask_b32 v12, v23, v12, vcc\n v_cndmask_b32 v13, v24, v13, vcc\n"
                  "v_cndmask_b32 v14, v25, v14, vcc\n v_cndmask_b32 v15, v26, v15, vcc\n v_cndmask_b32 v16, v27, v16, vcc\n v_cndmask_b32 v17, v28, v17, vcc\n")
KERNEL(k_cmp_cnd3, "v_cmp_ge_f32 vcc, v10, v21\n v_cndmask_b32 v11, v22, v11, vcc\n v_cndmask_b32 v12, v23, v12, vcc\n v_cndmask_b32 v13, v24, v13, vcc\n"
                   "v_cmp_ge_f32 vcc, v14, v25\n v_cndmask_b32 v15, v26, v15, vcc\n v_cndmask_b32 v16, v27, v16, vcc\n v_cndmask_b32 v17, v28, v17, vcc\n")

typedef void (*kern_t)(unsigned *, unsigned);

static size_t g_lds = 0; // dynamic LDS per block, only to cap the number of resident blocks per CU

static double time_ms(kern_t k, unsigned *d_out, int blocks)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), g_lds, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), g_lds, 0, d_out, 1u);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 3;
}

int main()
{
    const int blocks = 256 * 8;
    unsigned *d_out;
    CHECK(hipMalloc(&d_out, sizeof(unsigned) * blocks * 256));
    struct { const char *name; kern_t k; } tests[] = {
        {"v_exp_f32 (anchor, 8 cycles)", k_exp},
        {"v_xor_b32 vgpr, other bank", k_xor_diff}, {"v_xor_b32 vgpr, same bank", k_xor_same},
        {"v_xor_b32 sgpr", k_xor_sgpr}, {"v_xor_b32 inline constant", k_xor_cnst},
        {"v_mul_f32 vgpr, other bank", k_mul_diff}, {"v_mul_f32 vgpr, same bank", k_mul_same}, {"v_mul_f32 sgpr", k_mul_sgpr},
        {"v_add_f32 vgpr, other bank", k_add_diff}, {"v_add_f32 vgpr, same bank", k_add_same},
        {"v_add_u32 vgpr, other bank", k_addu_diff}, {"v_add_u32 vgpr, same bank", k_addu_same},
        {"v_fma_f32 three banks", k_fma_3banks}, {"v_fma_f32 two sources in one bank", k_fma_conflict},
        {"v_xor_b32 32-bit literal", k_xor_lit}, {"v_sub_f32 32-bit literal", k_sub_lit},
        {"v_max_f32 vgpr", k_max_diff}, {"v_min_f32 vgpr", k_min_diff},
        {"v_mad_u64_u32 vgpr multiplier", k_mad_vgpr}, {"v_mad_u64_u32 sgpr multiplier", k_mad_sgpr},
        {"v_mul_hi_u32 vgpr", k_mulhi_vgpr}, {"v_mul_lo_u32 vgpr", k_mullo_vgpr},
        {"v_alignbit_b32 vgpr, vgpr, 9", k_alignbit_const}, {"v_perm_b32 vgpr selector", k_perm_vgpr},
        {"v_perm_b32 sgpr selector", k_perm_sgpr}, {"v_bfi_b32 vgpr", k_bfi_vgpr}, {"v_xad_u32 vgpr", k_xad_vgpr},
        {"v_add3_u32 vgpr", k_add3_vgpr},
        {"v_bitop3_b32 (xor3) vgpr", k_bitop3_vgpr}, {"v_bitop3_b32 (xor3) sgpr key", k_bitop3_sgpr},
        {"v_cmp_ge_f32 -> vcc", k_cmp_vcc}, {"v_cndmask_b32 (vcc)", k_cnd_vcc}, {"v_cmp + 3 v_cndmask (x2)", k_cmp_cnd3},
        {"v_exp_f32 (again)", k_exp},
    };
    const double per_simd = 8.0 * ITERS * 4 * 8; // waves * iterations * unroll * chains
    const double t_exp = time_ms(k_exp, d_out, blocks);
    const double ghz = per_simd * 8.0 / (t_exp * 1e-3) / 1e9;
    std::printf("clock implied by v_exp_f32 at 8 cycles: %.3f GHz\n", ghz);
    for (auto &t : tests) {
        const double ms = time_ms(t.k, d_out, blocks);
        std::printf("%-36s %8.3f ms  %5.2f cycles\n", t.name, ms, ms * 1e-3 * ghz * 1e9 / per_simd);
    }
    // occupancy: the same loops with 8, 6, 5, 4, 3, 2, 1 resident blocks (x4 waves) per CU
    std::printf("\nwaves per SIMD    v_xor(vgpr)  v_xor(sgpr)  v_mad_u64  v_exp   [cycles per instruction]\n");
    const int per_cu[] = {8, 6, 5, 4, 3, 2, 1};
    for (int n : per_cu) {
        g_lds = n == 8 ? 0 : (size_t)(160 * 1024 / n) - 1024;
        if (g_lds > 64 * 1024) {
            kern_t ks[] = {k_xor_diff, k_xor_sgpr, k_mad_vgpr, k_exp};
            for (kern_t k : ks) CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds));
        }
        const double c = 1e-3 * ghz * 1e9 / per_simd;
        std::printf("%8d          %8.2f     %8.2f   %8.2f  %6.2f\n", n, time_ms(k_xor_diff, d_out, blocks) * c,
                    time_ms(k_xor_sgpr, d_out, blocks) * c, time_ms(k_mad_vgpr, d_out, blocks) * c, time_ms(k_exp, d_out, blocks) * c);
    }
    CHECK(hipFree(d_out));
    return 0;
}
