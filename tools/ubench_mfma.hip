// tools/ubench_mfma.hip -- what the matrix pipe costs beside vector work on MI355X (for chain_lag_kernel.hpp):
//   mfma16 / mfma32 : back-to-back v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 on NACC independent accumulators,
//                     W waves per SIMD -> SIMD cycles per MFMA (s_memtime around the loop, wave 0 of each block)
//   mix             : each wave alternates a burst of B MFMAs with V dependent-free v_fma_f32: the cost of an MFMA next to
//                     vector instructions of the SAME wave and of the other waves of the SIMD
//   VOP             : the vector instruction of the mix -- 0 v_fma_f32, 1 v_xor_b32, 2 v_mad_u64_u32, 3 v_exp_f32 (does the
//                     matrix pipe overlap with integer / transcendental vector work where it does not with FP32 FMAs?)
// Build: make ubench (tools/bin/ubench_mfma).  Prints one JSON line per case.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess) {                                                   \
            std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
            std::exit(1);                                                        \
        }                                                                        \
    } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// SHAPE 16: 16x16x32 (4 acc regs), 32: 32x32x16 (16 acc regs).  NACC accumulators, BURST MFMAs then NV v_fma per iteration.
template <int SHAPE, int NACC, int BURST, int NV, int VOP = 0>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float seed)
{
    half8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(seed + threadIdx.x + i), b[i] = (_Float16)(seed * 2 + i);
    f4 c16[NACC];
    f16v c32[NACC];
    for (int n = 0; n < NACC; ++n) {
        for (int i = 0; i < 4; ++i) c16[n][i] = 0.0f;
        for (int i = 0; i < 16; ++i) c32[n][i] = 0.0f;
    }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    unsigned long long w[8];
    for (int i = 0; i < 8; ++i) w[i] = (unsigned long long)(seed * 1000) + i + threadIdx.x;
    const unsigned mulc = 0xD2511F53u + (unsigned)seed;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < BURST; ++m) {
            if constexpr (SHAPE == 16) c16[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c16[m % NACC], 0, 0, 0);
            else c32[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c32[m % NACC], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < NV; ++m) {
            if constexpr (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[m % 8]) : "v"(seed));
            else if constexpr (VOP == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[m % 8]) : "v"(seed));
            else if constexpr (VOP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[m % 8]) : "v"(mulc), "v"(v[m % 8]) : "vcc");
            else asm volatile("v_exp_f32 %0, %0" : "+v"(v[m % 8]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int n = 0; n < NACC; ++n) {
        for (int i = 0; i < 4; ++i) s += c16[n][i];
        for (int i = 0; i < 16; ++i) s += c32[n][i];
    }
    for (int i = 0; i < 8; ++i) s += v[i] + (float)w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int NACC, int BURST, int NV, int VOP = 0>
void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 2000;
    float *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<SHAPE, NACC, BURST, NV, VOP>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto x : h) mean += (double)x;
    mean /= blocks;
    // SIMD cycles per iteration of one wave = elapsed / iters; per SIMD: W waves share it
    std::printf("{\"case\": \"%s\", \"shape\": %d, \"acc\": %d, \"burst\": %d, \"valu\": %d, \"waves_per_simd\": %d, "
                "\"cycles_per_iter_per_wave\": %.1f, \"simd_cycles_per_iter\": %.1f}\n",
                name, SHAPE, NACC, BURST, NV, waves_per_simd, mean / iters, mean / iters / waves_per_simd);
    CHECK(hipFree(out));
    CHECK(hipFree(cyc));
}

int main()
{
    for (int w : {1, 3}) {
        run<16, 4, 32, 0>("mfma16 4acc", w);
        run<16, 1, 32, 0>("mfma16 1acc", w);
        run<32, 2, 16, 0>("mfma32 2acc", w);
        run<16, 4, 0, 256>("valu only", w);
        run<16, 4, 32, 256>("burst32 + 256 fma", w);
        run<16, 4, 8, 64>("burst8 + 64 fma", w);
        run<16, 4, 1, 8>("1 mfma + 8 fma", w);
        run<32, 2, 16, 256>("mfma32 burst16 + 256 fma", w);
        run<16, 4, 0, 256, 1>("xor only", w);
        run<16, 4, 32, 256, 1>("burst32 + 256 xor", w);
        run<16, 4, 1, 8, 1>("1 mfma + 8 xor", w);
        run<16, 4, 0, 128, 2>("mad_u64 only", w);
        run<16, 4, 32, 128, 2>("burst32 + 128 mad_u64", w);
        run<16, 4, 1, 4, 2>("1 mfma + 4 mad_u64", w);
        run<16, 4, 0, 64, 3>("exp only", w);
        run<16, 4, 32, 64, 3>("burst32 + 64 exp", w);
        run<16, 4, 1, 2, 3>("1 mfma + 2 exp", w);
    }
    return 0;
}
