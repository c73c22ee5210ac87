// tools/ubench.hip -- VALU issue-rate microbenchmark for gfx950: how many
// lane-operations per second the chip sustains for the instruction classes the
// MH kernel is made of (FMA, 32x32->64 multiply, xor, transcendentals).  The
// numbers calibrate the VALU roofline in DESIGN.md.  Build: make ubench.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e = (x);                                                             \
        if (e != hipSuccess) {                                                          \
            std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);        \
            std::exit(1);                                                               \
        }                                                                               \
    } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

enum { OP_FMA, OP_MAD64, OP_MULHI, OP_MULLO, OP_XOR, OP_EXP, OP_RCP, OP_SIN, OP_LOG, OP_SQRT, OP_ADD64, OP_CVT, OP_MAD64ACC, OP_N };
static const char *names[OP_N] = {"v_fma_f32", "v_mad_u64_u32", "v_mul_hi_u32", "v_mul_lo_u32", "v_xor_b32",
                                  "v_exp_f32", "v_rcp_f32", "v_sin_f32", "v_log_f32", "v_sqrt_f32", "v_add_f64",
                                  "v_cvt_f32_u32", "v_mad_u64_u32 (acc)"};

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float seed)
{
    float f[ILP];
    unsigned u[ILP];
    double d[ILP];
    unsigned long long a64[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
        a64[i] = threadIdx.x * 0x9E3779B97F4A7C15ull + i;
        f[i] = seed + threadIdx.x * 1e-3f + i;
        u[i] = threadIdx.x * 2654435761u + i;
        d[i] = f[i];
    }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if constexpr (OP == OP_FMA) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
            if constexpr (OP == OP_MAD64) {
                unsigned long long p = (unsigned long long)u[i] * 0xD2511F53u;
                u[i] = (unsigned)(p >> 32) ^ (unsigned)p;
            }
            if constexpr (OP == OP_MULHI) u[i] = __umulhi(u[i], 0xD2511F53u) + 1u;
            if constexpr (OP == OP_MULLO) u[i] = u[i] * 0xD2511F53u + 1u;
            if constexpr (OP == OP_XOR) u[i] = (u[i] ^ 0x9E3779B9u) + 1u;
            if constexpr (OP == OP_EXP) f[i] = __builtin_amdgcn_exp2f(f[i]);
            if constexpr (OP == OP_RCP) f[i] = __builtin_amdgcn_rcpf(f[i]);
            if constexpr (OP == OP_SIN) f[i] = __builtin_amdgcn_sinf(f[i]);
            if constexpr (OP == OP_LOG) f[i] = __builtin_amdgcn_logf(f[i]);
            if constexpr (OP == OP_SQRT) f[i] = __builtin_amdgcn_sqrtf(f[i]);
            if constexpr (OP == OP_ADD64) d[i] = d[i] + 1.5;
            // one v_mad_u64_u32 per op and nothing else: D = lo(D) * K + D
            if constexpr (OP == OP_MAD64ACC) a64[i] = (unsigned long long)(unsigned)a64[i] * 0xD2511F53u + a64[i];
            if constexpr (OP == OP_CVT) f[i] = (float)(__builtin_bit_cast(unsigned, f[i]));
        }
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc += f[i] + (float)u[i] + (float)d[i] + (float)a64[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
void run(float *d_out, int blocks)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    // the XOR / MULLO / MULHI bodies carry one extra full-rate add per op (keeps the chain non-trivial)
    const double ops = (double)blocks * 256 * ITERS * ILP;
    std::printf("%-16s %8.3f ms  %7.2f Tlane-op/s  (%.2f lane-ops/clk/CU at 2.4 GHz)\n", names[OP], ms, ops / ms * 1e-9,
                ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main()
{
    const int blocks = 256 * 8 * 4;
    float *d_out;
    CHECK(hipMalloc(&d_out, sizeof(float) * blocks * 256));
    run<OP_FMA>(d_out, blocks);
    run<OP_MAD64>(d_out, blocks);
    run<OP_MULHI>(d_out, blocks);
    run<OP_MULLO>(d_out, blocks);
    run<OP_XOR>(d_out, blocks);
    run<OP_EXP>(d_out, blocks);
    run<OP_RCP>(d_out, blocks);
    run<OP_SIN>(d_out, blocks);
    run<OP_LOG>(d_out, blocks);
    run<OP_SQRT>(d_out, blocks);
    run<OP_ADD64>(d_out, blocks);
    run<OP_CVT>(d_out, blocks);
    run<OP_MAD64ACC>(d_out, blocks);
    CHECK(hipFree(d_out));
    return 0;
}
