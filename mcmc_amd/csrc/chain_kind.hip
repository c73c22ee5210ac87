// chain_kind.hip -- kernel instantiations of chain_kernel.hpp for ONE target
// kind (compiled four times with -DMCMC_KIND=0..3 so the build parallelises).
#include <atomic>

#include "chain_lag_kernel.hpp"
#include "host_common.hpp"

using namespace mcmc;

#ifndef MCMC_KIND
#error "compile with -DMCMC_KIND=<mcmc_target_kind>"
#endif

namespace {

constexpr int NLAG_MAX = MCMC_MAX_LAGS;

template <auto Kernel>
void launch_lag(const ChainParams &p, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL(Kernel, grid, dim3(LAG_THREADS), 0, s, p);
}

// 49 lags: FP32 takes chain_lag_kernel (lag products on the matrix cores) unless MCMC_LAG_MFMA=0 asks for the VALU form
template <typename Real, int KIND, int RULE, int MATH, bool INJ>
void launch_with_lags(const ChainParams &p, dim3 grid, hipStream_t s)
{
    if constexpr (sizeof(Real) == 4) {
        if (p.lag_mfma) { // `grid` is lag_grid_x(n_chains) x n_coords (chain.hip)
            if (p.samples != nullptr || p.accept_bits != nullptr) launch_lag<chain_lag_kernel<KIND, RULE, MATH, INJ, 1>>(p, grid, s);
            else launch_lag<chain_lag_kernel<KIND, RULE, MATH, INJ, 0>>(p, grid, s);
            return;
        }
    }
    launch_chain<Real, KIND, RULE, MATH, NLAG_MAX, INJ>(p, grid, s);
}

template <typename Real, int KIND, int RULE, int MATH>
void pick_lag_inj(const ChainParams &p, dim3 grid, hipStream_t s, bool inj)
{
    if (p.n_lags == 0) {
        if (inj) launch_chain<Real, KIND, RULE, MATH, 0, true>(p, grid, s);
        else launch_chain<Real, KIND, RULE, MATH, 0, false>(p, grid, s);
    } else {
        if (inj) launch_with_lags<Real, KIND, RULE, MATH, true>(p, grid, s);
        else launch_with_lags<Real, KIND, RULE, MATH, false>(p, grid, s);
    }
}

// diagnostic: the stamped instantiations (mcmc_chain_clock_probe)
template <int KIND>
int pick_stamped(const mcmc_chain_args *a, const ChainParams &p, dim3 grid, hipStream_t s)
{
    if (a->precision != 32 || a->math_mode != MCMC_MATH_FAST || a->inj_normals || a->samples || a->accept_bits)
        return set_error(MCMC_EINVAL, "the clock probe runs the FP32 FAST-math Philox kernel without write-back");
    if (a->accept_rule == MCMC_ACCEPT_CROSS) {
        if (p.n_lags == 0) launch_chain_stamped<KIND, MCMC_ACCEPT_CROSS, 0>(p, grid, s);
        else if (p.lag_mfma) launch_lag<chain_lag_kernel<KIND, MCMC_ACCEPT_CROSS, MCMC_MATH_FAST, false, 0, true>>(p, grid, s);
        else launch_chain_stamped<KIND, MCMC_ACCEPT_CROSS, NLAG_MAX>(p, grid, s);
    } else if (a->accept_rule == MCMC_ACCEPT_RECIP) {
        if (p.n_lags == 0) launch_chain_stamped<KIND, MCMC_ACCEPT_RECIP, 0>(p, grid, s);
        else if (p.lag_mfma) launch_lag<chain_lag_kernel<KIND, MCMC_ACCEPT_RECIP, MCMC_MATH_FAST, false, 0, true>>(p, grid, s);
        else launch_chain_stamped<KIND, MCMC_ACCEPT_RECIP, NLAG_MAX>(p, grid, s);
    } else {
        return set_error(MCMC_EINVAL, "the clock probe is instantiated for MCMC_ACCEPT_CROSS and MCMC_ACCEPT_RECIP");
    }
    return MCMC_OK;
}

template <int KIND>
int pick_rule(const mcmc_chain_args *a, const ChainParams &p, dim3 grid, hipStream_t s)
{
    if (p.stamps != nullptr) return pick_stamped<KIND>(a, p, grid, s);
    const bool inj = a->inj_normals != nullptr;
    if (a->precision == 64) {
        if (a->accept_rule != MCMC_ACCEPT_SHORTCIRCUIT)
            return set_error(MCMC_EINVAL, "precision 64 implements accept_rule MCMC_ACCEPT_SHORTCIRCUIT only");
        pick_lag_inj<double, KIND, MCMC_ACCEPT_SHORTCIRCUIT, MCMC_MATH_PRECISE>(p, grid, s, inj);
        return MCMC_OK;
    }
    if (a->precision != 32) return set_error(MCMC_EINVAL, "precision must be 32 or 64");
    if (a->math_mode == MCMC_MATH_PRECISE) {
        if (a->accept_rule != MCMC_ACCEPT_RECIP)
            return set_error(MCMC_EINVAL, "precise math implements accept_rule MCMC_ACCEPT_RECIP only (32-bit)");
        pick_lag_inj<float, KIND, MCMC_ACCEPT_RECIP, MCMC_MATH_PRECISE>(p, grid, s, inj);
        return MCMC_OK;
    }
    if (a->math_mode != MCMC_MATH_FAST) return set_error(MCMC_EINVAL, "bad math_mode");
    switch (a->accept_rule) {
    case MCMC_ACCEPT_RECIP: pick_lag_inj<float, KIND, MCMC_ACCEPT_RECIP, MCMC_MATH_FAST>(p, grid, s, inj); break;
    case MCMC_ACCEPT_NODIV: pick_lag_inj<float, KIND, MCMC_ACCEPT_NODIV, MCMC_MATH_FAST>(p, grid, s, inj); break;
    case MCMC_ACCEPT_CROSS: pick_lag_inj<float, KIND, MCMC_ACCEPT_CROSS, MCMC_MATH_FAST>(p, grid, s, inj); break;
    case MCMC_ACCEPT_LOG: pick_lag_inj<float, KIND, MCMC_ACCEPT_LOG, MCMC_MATH_FAST>(p, grid, s, inj); break;
    default: return set_error(MCMC_EINVAL, "32-bit accept_rule must be RECIP, NODIV, LOG or CROSS");
    }
    return MCMC_OK;
}

} // namespace

#define MCMC_CAT2(a, b) a##b
#define MCMC_CAT(a, b) MCMC_CAT2(a, b)

namespace mcmc {
int MCMC_CAT(chain_dispatch_kind_, MCMC_KIND)(const mcmc_chain_args *a, const ChainParams &p, dim3 grid, hipStream_t s)
{
    return pick_rule<MCMC_KIND>(a, p, grid, s);
}
} // namespace mcmc
