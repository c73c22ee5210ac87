/*
 * mcmc_distributions.h -- the reference's target-density callables, kept under
 * their driver-facing names so existing integrand code keeps compiling:
 *   xsection/include/Distributions.cuh:7-28  (float,  host + device)
 *   stepsize/include/metropolis.hpp:9-17     (double, host)
 * Plain C++ / HIP inlines; with hipcc they are __host__ __device__.
 * The arithmetic is the written expression (no fast-math): these are the
 * "precise" forms the parity tests compare against.
 */
#ifndef MCMC_DISTRIBUTIONS_H
#define MCMC_DISTRIBUTIONS_H

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MCMC_HD __host__ __device__ inline
#else
#define MCMC_HD inline
#endif

/* Distributions.cuh:7-10 */
MCMC_HD float WoodsSaxon(float r, float c, float a)
{
    if (r < 0.0f) return 0.0f;
    return 1 / (1 + expf((r - c) / a));
}

/* Distributions.cuh:13-16 */
MCMC_HD float R2WoodsSaxon(float r, float c, float a)
{
    if (r < 0.0f) return 0.0f;
    return r * r / (1 + expf((r - c) / a));
}

/* Distributions.cuh:19-22 */
MCMC_HD float Gaussian(float r, float variance)
{
    if (r < 0.0f) return 0.0f;
    return expf(-r * r / (2 * variance));
}

/* Distributions.cuh:25-28 */
MCMC_HD float R2Gaussian(float r, float variance)
{
    if (r < 0.0f) return 0.0f;
    return r * r * expf(-r * r / (2 * variance));
}

/* metropolis.hpp:9-12 */
MCMC_HD double R2WoodsSaxon(double r, double c, double a)
{
    if (r < 0.0) return 0.0;
    return r * r / (1 + exp((r - c) / a));
}

/* metropolis.hpp:14-17 */
MCMC_HD double R2Gaussian(double r, double variance)
{
    if (r < 0.0) return 0.0;
    return r * r * exp(-r * r / (2 * variance));
}

#endif /* MCMC_DISTRIBUTIONS_H */
