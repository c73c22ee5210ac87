// host_common.hpp -- error plumbing shared by the C-ABI translation units.
// The reference aborts the process on any CUDA error (Common.cuh:23-29,
// cudaCheckError -> exit(1)); the library returns codes instead.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mcmc_hip.h"

namespace mcmc {

int set_error(int code, const char *msg);
int set_hip_error(hipError_t e, const char *file, int line);

// Layer B / C entry points switch to their handle's (shards') device; this puts the calling thread's current device
// back when they return, so a caller that works on another GPU between calls (torch, a second handle) is not disturbed.
struct DeviceGuard {
    int prev = -1;
    DeviceGuard()
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

} // namespace mcmc

#define MCMC_HIP_CHECK(expr)                                                   \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) return ::mcmc::set_hip_error(_e, __FILE__, __LINE__); \
    } while (0)
