// tools/ubench2.hip -- issue cost (SIMD cycles per wave64 instruction) of the exact instructions in the chain
// kernel's hot loop, each pinned with inline asm: 8 independent dependency chains per lane, 8 waves per SIMD,
// so the figure is issue throughput, not latency.  cycles = SIMDs * clock * time / wave-instructions, with the
// clock taken from the v_exp_f32 row (quarter rate = 8 cycles).  Build: make ubench.
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

constexpr int ITERS = 8192;
constexpr int ILP = 8;
constexpr int UNROLL = 4;

enum { I_XOR, I_ADDF, I_MULF, I_FMA3, I_FMAC, I_CNDMASK, I_ALIGNBIT, I_PERM, I_MAD64, I_MAD64_XOR2, I_MAX, I_CMP, I_EXP, I_SUBF, I_CMP_CND3, I_XOR_AGAIN, I_N };
static const char *names[I_N] = {"v_xor_b32_e32", "v_add_f32_e32", "v_mul_f32_e32", "v_fma_f32 (3 vgpr)", "v_fmac_f32_e32",
                                 "v_cndmask_b32_e32", "v_alignbit_b32", "v_perm_b32", "v_mad_u64_u32 (+0)",
                                 "mad_u64 + 2 xor (Philox half round)", "v_max_f32_e32", "v_cmp_ge_f32 (vcc)", "v_exp_f32_e32",
                                 "v_sub_f32_e32", "v_cmp + 3 v_cndmask (accept/select)", "v_xor_b32_e32 (again, last)"};
static const int insts_per_op[I_N] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 3, 1, 1, 1, 1, 4, 1};

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed)
{
    unsigned a[ILP], b[ILP];
    unsigned long long w[ILP];
    unsigned lo32[ILP], hi32[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
        lo32[i] = threadIdx.x + i;
        hi32[i] = threadIdx.x * 3u + i;
        a[i] = seed + threadIdx.x * 2654435761u + i * 7919u;
        b[i] = a[i] * 31u + 1u;
        w[i] = a[i];
    }
    const unsigned c = seed * 3u + 0x3f800000u, m = 0xD2511F53u;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < UNROLL; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if constexpr (OP == I_XOR_AGAIN) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if constexpr (OP == I_CMP_CND3) {
                    asm volatile("v_cmp_ge_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %1, %0, vcc\n\t"
                                 "v_cndmask_b32_e32 %2, %3, %2, vcc\n\tv_cndmask_b32_e32 %3, %1, %3, vcc"
                                 : "+v"(a[i]), "+v"(b[i]), "+v"(lo32[i]), "+v"(hi32[i])
                                 :
                                 : "vcc");
                }
                if constexpr (OP == I_XOR) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if constexpr (OP == I_ADDF) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if constexpr (OP == I_SUBF) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if constexpr (OP == I_MULF) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if constexpr (OP == I_MAX) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if constexpr (OP == I_FMA3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c));
                if constexpr (OP == I_FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));
                if constexpr (OP == I_CNDMASK) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b[i]) : );
                if constexpr (OP == I_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a[i]) : "v"(c));
                if constexpr (OP == I_PERM) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b[i]), "s"(0x04000201u));
                if constexpr (OP == I_MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "s"(m) : "vcc");
                if constexpr (OP == I_MAD64_XOR2) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\tv_xor_b32_e32 %1, %3, %1\n\tv_xor_b32_e32 %1, %4, %1"
                                 : "=&v"(w[i]), "+v"(a[i])
                                 : "s"(m), "v"(b[i]), "v"(c)
                                 : "vcc");
                }
                if constexpr (OP == I_CMP) asm volatile("v_cmp_ge_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
                if constexpr (OP == I_EXP) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(a[i]));
            }
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc += a[i] + (unsigned)w[i] + (unsigned)(w[i] >> 32) + lo32[i] + hi32[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static double ms_of[I_N];

template <int OP>
void run(unsigned *d_out, int blocks)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    ms_of[OP] = ms / 5;
}

int main()
{
    const int blocks = 256 * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD, one round
    unsigned *d_out;
    CHECK(hipMalloc(&d_out, sizeof(unsigned) * blocks * 256));
    run<I_XOR>(d_out, blocks);
    run<I_ADDF>(d_out, blocks);
    run<I_SUBF>(d_out, blocks);
    run<I_MULF>(d_out, blocks);
    run<I_FMA3>(d_out, blocks);
    run<I_FMAC>(d_out, blocks);
    run<I_CNDMASK>(d_out, blocks);
    run<I_ALIGNBIT>(d_out, blocks);
    run<I_PERM>(d_out, blocks);
    run<I_MAD64>(d_out, blocks);
    run<I_MAD64_XOR2>(d_out, blocks);
    run<I_MAX>(d_out, blocks);
    run<I_CMP>(d_out, blocks);
    run<I_EXP>(d_out, blocks);
    run<I_CMP_CND3>(d_out, blocks);
    run<I_XOR_AGAIN>(d_out, blocks);
    // wave-instructions per SIMD: 8 waves * ITERS * UNROLL * ILP * insts
    const double per_simd = 8.0 * ITERS * UNROLL * ILP;
    const double ghz = per_simd * 8.0 / (ms_of[I_EXP] * 1e-3) / 1e9; // anchor: v_exp_f32 is quarter rate, 8 cycles
    std::printf("clock implied by v_exp_f32 at 8 cycles: %.3f GHz\n", ghz);
    for (int op = 0; op < I_N; ++op) {
        const double cyc = ms_of[op] * 1e-3 * ghz * 1e9 / per_simd;
        std::printf("%-38s %8.3f ms  %6.2f cycles per op (%d instruction%s)\n", names[op], ms_of[op], cyc, insts_per_op[op],
                    insts_per_op[op] > 1 ? "s" : "");
    }
    CHECK(hipFree(d_out));
    return 0;
}
