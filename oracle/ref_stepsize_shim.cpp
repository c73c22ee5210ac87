// ref_stepsize_shim.cpp -- builds the REFERENCE's own stepsize headers, where
// they lie under /root/reference/stepsize/include, into oracle/_ref/ so the
// CPU restatement (mcmc_oracle.c) can be pinned against the real thing and
// golden vectors generated (tests/golden/make_golden.py).
//
// TEST INFRASTRUCTURE ONLY.  No reference source is copied: metropolis.hpp and
// statistics.hpp are #included unmodified at build time (oracle/Makefile adds
// -I/root/reference/stepsize/include).  The only substitution is the random
// source: random.hpp seeds two thread-local mt19937 from the clock
// (random.hpp:8,15), which cannot be reproduced, so its include guard is
// pre-defined and RandomNormal/RandomUniform are supplied here, reading
// injected draws.  RandomNormal(0,1) -> 0 + 1*n and RandomUniform(0,1) ->
// 0 + (1-0)*u are exact, so the chain sees the injected values bit for bit.
#define RANDOM_H
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

static const double *g_normals = nullptr;
static const double *g_uniforms = nullptr;
static size_t g_ni = 0, g_ui = 0;

double RandomUniform(double min, double max) { return min + (max - min) * g_uniforms[g_ui++]; }
double RandomNormal(double mean, double stdev) { return mean + stdev * g_normals[g_ni++]; }

#include "metropolis.hpp"
#include "statistics.hpp"

extern "C" {

// MetropolisHastingsAlgorithm (metropolis.hpp:28) with target PDF() = r^2 WS(2.0, 0.54).
__attribute__((visibility("default")))
double ref_MetropolisHastingsAlgorithm(int iterations, double stepSize, const double *normals,
                                       const double *uniforms, double *samples_out, int *n_uniform_used)
{
    g_normals = normals; g_uniforms = uniforms; g_ni = 0; g_ui = 0;
    auto res = MetropolisHastingsAlgorithm(iterations, stepSize);
    for (int i = 0; i < iterations; ++i) samples_out[i] = res.second[i];
    if (n_uniform_used) *n_uniform_used = (int)g_ui;
    return res.first;
}

__attribute__((visibility("default")))
double ref_R2WoodsSaxon(double r, double c, double a) { return R2WoodsSaxon(r, c, a); }
__attribute__((visibility("default")))
double ref_R2Gaussian(double r, double v) { return R2Gaussian(r, v); }
__attribute__((visibility("default")))
double ref_PDF(double r) { return PDF(r); }

__attribute__((visibility("default")))
double ref_Average(const double *x, int n) { return Average(std::vector<double>(x, x + n)); }
__attribute__((visibility("default")))
double ref_Variance(const double *x, int n) { return Variance(std::vector<double>(x, x + n)); }
__attribute__((visibility("default")))
double ref_AutoCorrelation(const double *x, int n, int lag) { return AutoCorrelation(std::vector<double>(x, x + n), lag); }

}
