// tools/dist_probe.hip -- one-kernel device probe of include/mcmc_distributions.h: the four FP32 callables of
// xsection/include/Distributions.cuh:7-28 evaluated ON THE DEVICE over a grid of radii (and on the host, same
// translation unit, through the __host__ side of the same inlines).  Prints "r c|variance device-bits host-bits" rows
// per function; tests/test_distributions_header.py compares them with the oracle.  Build: make probes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "mcmc_distributions.h"

constexpr int NR = 96;

__global__ void probe(const float *r, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NR) return;
    out[0 * NR + i] = WoodsSaxon(r[i], 2.0f, 0.54f);
    out[1 * NR + i] = R2WoodsSaxon(r[i], 2.0f, 0.54f);
    out[2 * NR + i] = Gaussian(r[i], 1.5f);
    out[3 * NR + i] = R2Gaussian(r[i], 1.0f);
    out[4 * NR + i] = R2WoodsSaxon(r[i], 6.62f, 0.546f);
}

static unsigned bits(float f)
{
    unsigned u;
    std::memcpy(&u, &f, 4);
    return u;
}

int main()
{
    std::vector<float> r(NR), dev(5 * NR);
    for (int i = 0; i < NR; ++i) r[i] = -0.5f + 0.171f * i; // -0.5 .. 15.7: r < 0, the bulk, and the expf-overflow tail
    r[NR - 1] = 200.0f;                                      // expf((r - c)/a) = inf -> 1/(1+inf) = 0 (must stay benign)
    float *d_r, *d_o;
    if (hipMalloc(&d_r, NR * 4) != hipSuccess || hipMalloc(&d_o, 5 * NR * 4) != hipSuccess) return 2;
    if (hipMemcpy(d_r, r.data(), NR * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
    hipLaunchKernelGGL(probe, dim3(1), dim3(128), 0, 0, d_r, d_o);
    if (hipMemcpy(dev.data(), d_o, 5 * NR * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const char *names[5] = {"WoodsSaxon_2_0.54", "R2WoodsSaxon_2_0.54", "Gaussian_1.5", "R2Gaussian_1", "R2WoodsSaxon_6.62_0.546"};
    for (int f = 0; f < 5; ++f)
        for (int i = 0; i < NR; ++i) {
            const float h = f == 0 ? WoodsSaxon(r[i], 2.0f, 0.54f) : f == 1 ? R2WoodsSaxon(r[i], 2.0f, 0.54f)
                            : f == 2 ? Gaussian(r[i], 1.5f) : f == 3 ? R2Gaussian(r[i], 1.0f) : R2WoodsSaxon(r[i], 6.62f, 0.546f);
            std::printf("%s %08x %08x %08x\n", names[f], bits(r[i]), bits(dev[f * NR + i]), bits(h));
        }
    return 0;
}
