// tools/ubench_store.hip -- streaming-write ceilings of one MI355X by store width and pattern, to read the sample
// write-back kernels against (chain STORE, stress FULL write 4 B per lane and instruction: 256 B per wave-store).
//   width   : bytes per lane and store instruction (4 / 8 / 16): a wave-store covers 256 / 512 / 1024 contiguous bytes
//   pattern : "stream" = a wave walks its own contiguous span; "rows" = like the kernels: store k of a wave goes to row k
//             (row stride = n_lanes_total * 4 B), lane offset fixed
//   nt      : streaming (nontemporal) stores or plain ones
// Build: make tools/bin/ubench_store ; run: tools/bin/ubench_store
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Val;
template <> struct Val<float> { static __device__ float make(float v) { return v; } };
template <> struct Val<f32x2> { static __device__ f32x2 make(float v) { return f32x2{v, v}; } };
template <> struct Val<f32x4> { static __device__ f32x4 make(float v) { return f32x4{v, v, v, v}; } };

template <typename T, bool NT> __device__ __forceinline__ void st(T *p, T v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// rows: out[k][lane_global]; every thread writes `rows` values, one per row (the kernels' pattern)
template <typename T, bool NT> __global__ void rows_kernel(T *out, uint64_t n_per_row, int rows, float v)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_per_row) return;
    float x = v + (float)threadIdx.x;
    for (int k = 0; k < rows; ++k) {
        x = x * 1.0001f + 0.5f;
        st<T, NT>(out + (uint64_t)k * n_per_row + i, Val<T>::make(x));
    }
}

// stream: grid-stride over the whole buffer
template <typename T, bool NT> __global__ void stream_kernel(T *out, uint64_t n, float v)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    float x = v + (float)threadIdx.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        x = x * 1.0001f + 0.5f;
        st<T, NT>(out + i, Val<T>::make(x));
    }
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

template <typename T, bool NT> static void run(void *buf, uint64_t bytes, int rows)
{
    const uint64_t n = bytes / sizeof(T), n_per_row = n / rows;
    const unsigned grid_rows = (unsigned)((n_per_row + 255) / 256);
    const double ms_r = time_ms([&] { hipLaunchKernelGGL((rows_kernel<T, NT>), dim3(grid_rows), dim3(256), 0, 0, (T *)buf, n_per_row, rows, 1.0f); });
    const double ms_s = time_ms([&] { hipLaunchKernelGGL((stream_kernel<T, NT>), dim3(256 * 32), dim3(256), 0, 0, (T *)buf, n, 1.0f); });
    std::printf("{\"width_B\": %d, \"nt\": %s, \"bytes\": %llu, \"rows\": %d, \"rows_GBps\": %.1f, \"stream_GBps\": %.1f}\n", (int)sizeof(T),
                NT ? "true" : "false", (unsigned long long)bytes, rows, bytes / (ms_r * 1e-3) / 1e9, bytes / (ms_s * 1e-3) / 1e9);
}

int main()
{
    const uint64_t bytes = 6ull << 30;
    void *buf;
    CHECK(hipMalloc(&buf, bytes));
    for (int rows : {32, 96}) {
        run<float, true>(buf, bytes, rows);
        run<float, false>(buf, bytes, rows);
        run<f32x2, true>(buf, bytes, rows);
        run<f32x2, false>(buf, bytes, rows);
        run<f32x4, true>(buf, bytes, rows);
        run<f32x4, false>(buf, bytes, rows);
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipFree(buf));
    return 0;
}
