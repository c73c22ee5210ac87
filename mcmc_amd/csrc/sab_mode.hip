// sab_mode.hip -- kernel instantiations of sab_kernel (xs_kernel.hpp) for ONE mode (compiled four times with
// -DMCMC_SAB_MODE=1..4 so the build parallelises): 1 = S_AB product, 2 = black-disk hit counters, 3 = both,
// 4 = inelastic flag only.
#include <atomic>

#include "host_common.hpp"
#include "xs_kernel.hpp"

using namespace mcmc;

#ifndef MCMC_SAB_MODE
#error "compile with -DMCMC_SAB_MODE=<1..4>"
#endif

namespace {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per KERNEL and device instead of on every launch.  The cache is a
// static inside a template over the kernel itself: every sab_kernel instantiation has the same function-pointer TYPE, so a
// cache keyed on the type was one cache for all of them and skipped the attribute for every kernel after the first.
template <auto Kernel>
int ensure_dynamic_lds(size_t lds)
{
    static std::atomic<size_t> granted[64];
    if (lds <= 48 * 1024) return MCMC_OK;
    int dev = 0;
    MCMC_HIP_CHECK(hipGetDevice(&dev));
    const bool cached = dev >= 0 && dev < 64;
    if (cached && granted[dev].load(std::memory_order_acquire) >= lds) return MCMC_OK;
    MCMC_HIP_CHECK(hipFuncSetAttribute((const void *)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (cached) granted[dev].store(lds, std::memory_order_release);
    return MCMC_OK;
}

template <int G, int KIND, int MODE>
int launch_sab_g(const mcmc_xs_args *a, const XsParams &p, size_t lds, hipStream_t s)
{
    constexpr int EPB = XS_BLOCK / G;
    const dim3 grid((unsigned)((p.T + EPB - 1) / EPB));
    const dim3 block(XS_BLOCK);
    const bool inj = a->injected != nullptr;
#define MCMC_SAB_LAUNCH(RULE, MATH, INJ)                                  \
    do {                                                                  \
        constexpr auto k = sab_kernel<G, KIND, RULE, MATH, INJ, MODE>;    \
        const int rc_lds = ensure_dynamic_lds<k>(lds);                    \
        if (rc_lds != MCMC_OK) return rc_lds;                             \
        hipLaunchKernelGGL(k, grid, block, lds, s, p);                    \
    } while (0)
    if (a->math_mode == MCMC_MATH_PRECISE) {
        if (inj) MCMC_SAB_LAUNCH(MCMC_ACCEPT_RECIP, MCMC_MATH_PRECISE, true);
        else MCMC_SAB_LAUNCH(MCMC_ACCEPT_RECIP, MCMC_MATH_PRECISE, false);
    } else if (a->accept_rule == MCMC_ACCEPT_RECIP) {
        if (inj) MCMC_SAB_LAUNCH(MCMC_ACCEPT_RECIP, MCMC_MATH_FAST, true);
        else MCMC_SAB_LAUNCH(MCMC_ACCEPT_RECIP, MCMC_MATH_FAST, false);
    } else if (a->accept_rule == MCMC_ACCEPT_LOG) {
        if (inj) MCMC_SAB_LAUNCH(MCMC_ACCEPT_LOG, MCMC_MATH_FAST, true);
        else MCMC_SAB_LAUNCH(MCMC_ACCEPT_LOG, MCMC_MATH_FAST, false);
    } else {
        // MCMC_ACCEPT_NODIV, and MCMC_ACCEPT_CROSS (mcmc_hip.h: this kernel keeps one cached value per nucleon in
        // LDS, so the cross-multiplied quotient form is evaluated as the same decision p' >= U p)
        if (inj) MCMC_SAB_LAUNCH(MCMC_ACCEPT_NODIV, MCMC_MATH_FAST, true);
        else MCMC_SAB_LAUNCH(MCMC_ACCEPT_NODIV, MCMC_MATH_FAST, false);
    }
#undef MCMC_SAB_LAUNCH
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}

template <int G, int MODE>
int launch_sab_kind(const mcmc_xs_args *a, const XsParams &p, size_t lds, bool all_r2ws, hipStream_t s)
{
    if (all_r2ws) return launch_sab_g<G, MCMC_TARGET_R2WS, MODE>(a, p, lds, s);
    return launch_sab_g<G, -1, MODE>(a, p, lds, s);
}

} // namespace

#define MCMC_CAT2(a, b) a##b
#define MCMC_CAT(a, b) MCMC_CAT2(a, b)

namespace mcmc {
int MCMC_CAT(launch_sab_mode_, MCMC_SAB_MODE)(const mcmc_xs_args *a, const XsParams &p, int G, size_t lds, bool all_r2ws,
                                               hipStream_t s)
{
    switch (G) {
    case 4: return launch_sab_kind<4, MCMC_SAB_MODE>(a, p, lds, all_r2ws, s);
    case 16: return launch_sab_kind<16, MCMC_SAB_MODE>(a, p, lds, all_r2ws, s);
    default: return launch_sab_kind<64, MCMC_SAB_MODE>(a, p, lds, all_r2ws, s);
    }
}
} // namespace mcmc
