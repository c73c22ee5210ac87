// handle.hpp -- the Layer B handle, shared by capi.hip (one GPU) and multi.hip (several GPUs of one node).
#pragma once
#include "host_common.hpp"

struct mcmc_handle_s {
    mcmc_config cfg;
    int nA, nB, nN;
    uint64_t T;        // events on this handle's device
    hipStream_t stream;
    float *d_prev;
    mcmc_target *d_targets;
    float *d_result;   // [nN][T]
    double *d_means;   // [nN] + [4] hits
    void *d_ws;
    size_t ws_bytes;
    double *d_scan;    // per-launch results of mcmc_run_rms / mcmc_run_sab_scan
    size_t scan_len;
    uint64_t sweeps_done;
    int uniform_kind;
    char outDir[512];  // copies of cfg.outDir / cfg.outFile (the caller's strings need not outlive mcmc_create)
    char outFile[256];
};

namespace mcmc {

// mcmc_create for `events` events (cfg.totalBlocks * cfg.threadsPerBlock is the whole job's count; a shard owns a part)
int create_handle(const mcmc_config *cfg, uint64_t events, mcmc_handle *out);

inline mcmc_xs_args xs_args(mcmc_handle h, float stepsize)
{
    mcmc_xs_args a{};
    a.nNucleonsA = h->nA;
    a.nNucleonsB = h->nB;
    a.totalThreads = h->T;
    a.targets = h->d_targets;
    a.prevSample = h->d_prev;
    a.stepsize = stepsize;
    a.accept_rule = h->cfg.accept_rule;
    a.math_mode = h->cfg.math_mode;
    a.uniform_kind = h->uniform_kind;
    a.seed = h->cfg.seed;
    a.step_offset = h->sweeps_done;
    a.event_offset = h->cfg.event_offset;
    a.sigmaNN = h->cfg.sigmaNN;
    a.beta = h->cfg.beta;
    a.workspace = h->d_ws;
    a.workspace_bytes = h->ws_bytes;
    return a;
}

} // namespace mcmc
