// host_common.hpp -- error plumbing shared by the C-ABI translation units.
// The reference aborts the process on any CUDA error (Common.cuh:23-29,
// cudaCheckError -> exit(1)); the library returns codes instead.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mcmc_hip.h"

namespace mcmc {

int set_error(int code, const char *msg);
int set_hip_error(hipError_t e, const char *file, int line);

} // namespace mcmc

#define MCMC_HIP_CHECK(expr)                                                   \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) return ::mcmc::set_hip_error(_e, __FILE__, __LINE__); \
    } while (0)
