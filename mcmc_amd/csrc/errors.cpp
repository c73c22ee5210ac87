// errors.cpp -- thread-local last-error string of the C ABI.
#include <cstdio>
#include <cstring>

#include "host_common.hpp"

namespace {
thread_local char g_err[512] = "";
}

namespace mcmc {

int set_error(int code, const char *msg)
{
    std::snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int set_hip_error(hipError_t e, const char *file, int line)
{
    std::snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
    return (int)e;
}

} // namespace mcmc

extern "C" const char *mcmc_last_error(void) { return g_err; }
#ifndef MCMC_PHILOX_ROUNDS
#define MCMC_PHILOX_ROUNDS 10
#endif
#define MCMC_STR2(x) #x
#define MCMC_STR(x) MCMC_STR2(x)
extern "C" const char *mcmc_version(void) { return "mcmc_hip 0.2 (gfx950, philox4x32-" MCMC_STR(MCMC_PHILOX_ROUNDS) ")"; }
extern "C" int mcmc_philox_rounds(void) { return MCMC_PHILOX_ROUNDS; }
extern "C" int mcmc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
