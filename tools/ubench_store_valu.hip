// tools/ubench_store_valu.hip -- how much of a streaming store hides behind vector arithmetic on one MI355X: the "rows" pattern
// of tools/ubench_store.hip (store k of a lane goes to row k, 4 B per lane = 256 B per wave-store, the chain kernel's
// write-back) with WORK dependent-free v_fma_f32 per lane between two stores.  For every WORK: time with the stores, time of
// the same arithmetic without them, time of the stores alone (WORK = 0) -> the part of the shorter leg that did not overlap.
// Build: make tools/bin/ubench_store_valu ; run: tools/bin/ubench_store_valu [rows]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// stamps[2 b], [2 b + 1]: shader-clock cycles and 100 MHz ticks block b took (the shader clock the launch sustained)
template <int WORK, bool STORE> __global__ __launch_bounds__(256) void rows_kernel(float *out, uint64_t n_per_row, int rows, float a, float b,
                                                                                  uint64_t *stamps)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_per_row) return;
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x0 = a + (float)threadIdx.x, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f;
    for (int k = 0; k < rows; ++k) {
#pragma unroll
        for (int w = 0; w < WORK / 4; ++w) {
            x0 = __builtin_fmaf(x0, a, b);
            x1 = __builtin_fmaf(x1, a, b);
            x2 = __builtin_fmaf(x2, a, b);
            x3 = __builtin_fmaf(x3, a, b);
        }
        if constexpr (WORK == 0) x0 = __builtin_fmaf(x0, a, b);
        if constexpr (STORE) __builtin_nontemporal_store((x0 + x1) + (x2 + x3), out + (uint64_t)k * n_per_row + i);
    }
    if constexpr (!STORE) out[i] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) {
        stamps[2 * (blockIdx.x >> 6)] = __builtin_amdgcn_s_memtime() - c0;
        stamps[2 * (blockIdx.x >> 6) + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

static uint64_t *g_stamps = nullptr;
static size_t g_n_stamps = 0;
static double clock_ghz()
{
    std::vector<uint64_t> h(2 * g_n_stamps);
    hipMemcpy(h.data(), g_stamps, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
    std::vector<double> r;
    for (size_t b = 0; b < g_n_stamps; ++b)
        if (h[2 * b + 1] > 20) r.push_back(0.1 * (double)h[2 * b] / (double)h[2 * b + 1]);
    if (r.empty()) return 0.0;
    std::nth_element(r.begin(), r.begin() + r.size() / 2, r.end());
    return r[r.size() / 2];
}

template <typename F> static double time_ms(F launch, int reps = 5)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

static double g_store_only_ms = 0.0;

template <int WORK> static void run(float *buf, uint64_t n_per_row, int rows)
{
    const unsigned grid = (unsigned)((n_per_row + 255) / 256);
    const double bytes = (double)n_per_row * rows * 4.0;
    const double ms_s = time_ms([&] { hipLaunchKernelGGL((rows_kernel<WORK, true>), dim3(grid), dim3(256), 0, 0, buf, n_per_row, rows, 0.9999f, 0.5f, g_stamps); });
    const double ghz_s = clock_ghz();
    const double ms_c = time_ms([&] { hipLaunchKernelGGL((rows_kernel<WORK, false>), dim3(grid), dim3(256), 0, 0, buf, n_per_row, rows, 0.9999f, 0.5f, g_stamps); });
    const double ghz_c = clock_ghz();
    if (WORK == 0) g_store_only_ms = ms_s;
    const double lo = std::min(ms_c, g_store_only_ms), hi = std::max(ms_c, g_store_only_ms);
    std::printf("{\"fma_per_store\": %d, \"lanes\": %llu, \"rows\": %d, \"ms_with_stores\": %.3f, \"ms_arithmetic_only\": %.3f, \"ms_stores_only\": %.3f, "
                "\"GBps_with_stores\": %.1f, \"exposed_fraction_of_shorter_leg\": %.3f, \"clock_GHz_with_stores\": %.3f, \"clock_GHz_arithmetic_only\": %.3f, "
                "\"simd_cycles_per_wave_store_beyond_arithmetic\": %.1f}\n",
                WORK, (unsigned long long)n_per_row, rows, ms_s, ms_c, g_store_only_ms, bytes / (ms_s * 1e-3) / 1e9, lo > 0 ? (ms_s - hi) / lo : 0.0,
                ghz_s, ghz_c, (ms_s * ghz_s - ms_c * ghz_c) * 1e6 * 1024.0 / ((double)n_per_row * rows / 64.0));
}

int main(int argc, char **argv)
{
    const int rows = argc > 1 ? std::atoi(argv[1]) : 64;
    const uint64_t n_per_row = 3ull << 24; // the write-back bench's 2^24 chains x 3 coordinates
    float *buf;
    CHECK(hipMalloc((void **)&buf, n_per_row * rows * sizeof(float)));
    g_n_stamps = (size_t)((n_per_row + 255) / 256 / 64 + 1);
    CHECK(hipMalloc((void **)&g_stamps, 2 * g_n_stamps * sizeof(uint64_t)));
    CHECK(hipMemset(g_stamps, 0, 2 * g_n_stamps * sizeof(uint64_t)));
    run<0>(buf, n_per_row, rows);
    run<16>(buf, n_per_row, rows);
    run<24>(buf, n_per_row, rows);
    run<32>(buf, n_per_row, rows);
    run<40>(buf, n_per_row, rows);
    run<48>(buf, n_per_row, rows);
    run<64>(buf, n_per_row, rows);
    run<96>(buf, n_per_row, rows);
    CHECK(hipDeviceSynchronize());
    CHECK(hipFree(buf));
    return 0;
}
