// multi.hip -- Layer C of the C ABI: the two drivers on several GPUs of one node.
//
// The reference runs on one GPU and reduces on the host: Average over the threads of a run, then Average /
// Variance over runs (xsection/src/main.cu:221-223,253-258,359,384-389); the stepsize executable is one CPU chain
// (stepsize/src/main.cpp:17-30).  Here the events / chains are partitioned by GLOBAL id over the shards (one shard
// per GPU; contiguous, balanced: shard r owns [r*base + min(r, extra), ...), the first n % W shards one id more),
// every shard queues ALL its launches on its own stream from its own host thread, each launch leaving its sufficient
// statistics in one device vector, and ONE all-reduce(SUM, FP64) of that vector ends the batch:
//   * RCCL (librccl, loaded at run time; ncclCommInitAll + one grouped ncclAllReduce over the shards' streams), or
//   * the host (each shard copies its vector back, the host adds) -- the fallback when librccl does not load, and
//     the only choice when two shards share a device (a one-GPU box rehearsing the partition).
// The Philox counters carry the global id, so the union of the shards' events is exactly the event set of one GPU.
// xsection payloads are INTEGERS held as doubles (hit counts; S_AB and r^2 sums as fixed-point limbs,
// row_fixed_partial_kernel): their sum does not depend on the partition -- results are bit-identical for any
// number of shards.  stepsize payloads are the FP64 sufficient statistics of mcmc_chain_run (sums over chains:
// equal across partitions to FP64 rounding).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "handle.hpp"
#include "host_common.hpp"

using namespace mcmc;

namespace {

// ---------------------------------------------------------------------
// librccl at run time: the library stays loadable (and the one-GPU paths usable) where RCCL is not installed.
struct Rccl {
    void *so = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.so) break;
        }
        if (!r.so) return;
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.so, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.so, "ncclAllReduce");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.so, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.so, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.AllReduce && r.GroupStart && r.GroupEnd && r.GetErrorString;
    });
    return r;
}

int set_rccl_error(ncclResult_t e, const char *what)
{
    std::string msg = std::string("RCCL: ") + what + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "?");
    return set_error(MCMC_EINVAL, msg.c_str());
}

// contiguous, balanced partition of ids [0, n) (the same as mcmc_amd/dist.py: partition)
void partition(uint64_t n, int rank, int world, uint64_t *offset, uint64_t *count)
{
    const uint64_t base = n / (uint64_t)world, extra = n % (uint64_t)world;
    *count = base + ((uint64_t)rank < extra ? 1 : 0);
    *offset = (uint64_t)rank * base + ((uint64_t)rank < extra ? (uint64_t)rank : extra);
}

// Run fn(r) for every shard, one host thread per shard (launch queues fill in parallel); the first failure's code and
// message become the caller's (mcmc_last_error is thread-local).
template <typename F>
int for_each_shard(int W, F fn)
{
    std::vector<int> rc(W, MCMC_OK);
    std::vector<std::string> msg(W);
    auto body = [&](int r) {
        rc[r] = fn(r);
        if (rc[r] != MCMC_OK) msg[r] = mcmc_last_error();
    };
    if (W == 1) body(0);
    else {
        std::vector<std::thread> th;
        th.reserve(W);
        for (int r = 0; r < W; ++r) th.emplace_back(body, r);
        for (auto &t : th) t.join();
    }
    for (int r = 0; r < W; ++r)
        if (rc[r] != MCMC_OK) return set_error(rc[r], msg[r].c_str());
    return MCMC_OK;
}

// One device vector per shard, summed over the shards into `total` (host).  The streams have the launches that fill
// the vectors queued; this call orders the exchange behind them and waits.
struct ReducePeer {
    int device;
    hipStream_t stream;
    double *d_vec;
    ncclComm_t comm;
};

int all_reduce_sum(std::vector<ReducePeer> &peers, size_t len, bool use_rccl, std::vector<double> &total)
{
    const int W = (int)peers.size();
    total.assign(len, 0.0);
    if (use_rccl) {
        Rccl &R = rccl();
        ncclResult_t e = R.GroupStart();
        if (e != ncclSuccess) return set_rccl_error(e, "ncclGroupStart");
        for (int r = 0; r < W; ++r) {
            MCMC_HIP_CHECK(hipSetDevice(peers[r].device));
            e = R.AllReduce(peers[r].d_vec, peers[r].d_vec, len, ncclDouble, ncclSum, peers[r].comm, peers[r].stream);
            if (e != ncclSuccess) {
                (void)R.GroupEnd();
                return set_rccl_error(e, "ncclAllReduce");
            }
        }
        e = R.GroupEnd();
        if (e != ncclSuccess) return set_rccl_error(e, "ncclGroupEnd");
        MCMC_HIP_CHECK(hipSetDevice(peers[0].device));
        MCMC_HIP_CHECK(hipMemcpyAsync(total.data(), peers[0].d_vec, sizeof(double) * len, hipMemcpyDeviceToHost, peers[0].stream));
        for (int r = 0; r < W; ++r) {
            MCMC_HIP_CHECK(hipSetDevice(peers[r].device));
            MCMC_HIP_CHECK(hipStreamSynchronize(peers[r].stream));
        }
        return MCMC_OK;
    }
    std::vector<std::vector<double>> part(W, std::vector<double>(len));
    for (int r = 0; r < W; ++r) {
        MCMC_HIP_CHECK(hipSetDevice(peers[r].device));
        MCMC_HIP_CHECK(hipMemcpyAsync(part[r].data(), peers[r].d_vec, sizeof(double) * len, hipMemcpyDeviceToHost, peers[r].stream));
    }
    for (int r = 0; r < W; ++r) {
        MCMC_HIP_CHECK(hipSetDevice(peers[r].device));
        MCMC_HIP_CHECK(hipStreamSynchronize(peers[r].stream));
    }
    for (int r = 0; r < W; ++r) // shard order: fixed, so FP64 payloads are reproducible run to run
        for (size_t i = 0; i < len; ++i) total[i] += part[r][i];
    return MCMC_OK;
}

// AUTO -> RCCL when librccl loads, there is more than one shard and every shard has a device of its own
int resolve_reduce(int reduce, const int *devices, int W, bool *use_rccl)
{
    bool distinct = true;
    for (int i = 0; i < W; ++i)
        for (int j = i + 1; j < W; ++j)
            if (devices[i] == devices[j]) distinct = false;
    if (reduce == MCMC_REDUCE_HOST) *use_rccl = false;
    else if (reduce == MCMC_REDUCE_RCCL) {
        if (!rccl().ok) return set_error(MCMC_EINVAL, "MCMC_REDUCE_RCCL: librccl.so could not be loaded");
        if (!distinct) return set_error(MCMC_EINVAL, "MCMC_REDUCE_RCCL: two shards share a device (RCCL needs one device per rank)");
        *use_rccl = true;
    } else if (reduce == MCMC_REDUCE_AUTO) *use_rccl = rccl().ok && distinct && W > 1;
    else return set_error(MCMC_EINVAL, "reduce must be MCMC_REDUCE_AUTO, _HOST or _RCCL");
    return MCMC_OK;
}

int check_devices(const int *devices, int n_shards)
{
    if (!devices || n_shards <= 0) return set_error(MCMC_EINVAL, "need a device list with n_shards > 0 entries");
    if (n_shards > 64) return set_error(MCMC_EINVAL, "at most 64 shards");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_error(MCMC_ENODEV, "no HIP device");
    for (int i = 0; i < n_shards; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) return set_error(MCMC_EINVAL, "device ordinal out of range");
    return MCMC_OK;
}

} // namespace

// =====================================================================
// xsection on several GPUs.
struct mcmc_multi_s {
    mcmc_config cfg;
    int W = 0;
    bool use_rccl = false;
    uint64_t T_total = 0;
    int nA = 0, nB = 0, nN = 0;
    struct Shard {
        int device = 0;
        uint64_t offset = 0, count = 0;
        mcmc_handle h = nullptr;
        double *d_vec = nullptr; // this shard's payload
        size_t vec_len = 0;
        void *d_fixed_ws = nullptr;
        size_t fixed_ws_bytes = 0;
        ncclComm_t comm = nullptr;
    };
    std::vector<Shard> shards;
};

namespace {

int ensure_vec(mcmc_multi_s::Shard &s, size_t len)
{
    if (s.vec_len >= len) return MCMC_OK;
    MCMC_HIP_CHECK(hipStreamSynchronize(s.h->stream));
    if (s.d_vec) (void)hipFree(s.d_vec);
    s.d_vec = nullptr;
    s.vec_len = 0;
    MCMC_HIP_CHECK(hipMalloc((void **)&s.d_vec, sizeof(double) * len));
    s.vec_len = len;
    return MCMC_OK;
}

int reduce_multi(mcmc_multi m, size_t len, std::vector<double> &total)
{
    std::vector<ReducePeer> peers(m->W);
    for (int r = 0; r < m->W; ++r) peers[r] = ReducePeer{m->shards[r].device, m->shards[r].h->stream, m->shards[r].d_vec, m->shards[r].comm};
    return all_reduce_sum(peers, len, m->use_rccl, total);
}

int check_multi_scan(mcmc_multi m)
{
    if (!m) return set_error(MCMC_EINVAL, "null handle");
    if (m->cfg.totalRuns <= 0) return set_error(MCMC_EINVAL, "totalRuns must be > 0");
    if (m->cfg.samplesPerThread <= 0) return set_error(MCMC_EINVAL, "samplesPerThread must be > 0");
    return MCMC_OK;
}

} // namespace

extern "C" int mcmc_multi_create(const mcmc_config *cfg, const int *devices, int n_shards, int reduce, mcmc_multi *out)
{
    if (!cfg || !out) return set_error(MCMC_EINVAL, "null pointer");
    DeviceGuard restore_device;
    int rc = check_devices(devices, n_shards);
    if (rc) return rc;
    if (cfg->totalBlocks <= 0 || cfg->threadsPerBlock <= 0) return set_error(MCMC_EINVAL, "totalBlocks and threadsPerBlock must be > 0");
    const uint64_t T = (uint64_t)cfg->totalBlocks * (uint64_t)cfg->threadsPerBlock; // main.cu:89: the whole job's events
    if (T < (uint64_t)n_shards) return set_error(MCMC_EINVAL, "fewer events than shards");
    if (T > (1ull << 32)) return set_error(MCMC_EINVAL, "at most 2^32 events per job (exact fixed-point reduction)");
    bool use_rccl = false;
    if ((rc = resolve_reduce(reduce, devices, n_shards, &use_rccl)) != MCMC_OK) return rc;

    mcmc_multi m = new mcmc_multi_s();
    m->cfg = *cfg;
    m->cfg.nucleon_targets = nullptr; // read by the shards' create_handle below (from *cfg), not kept
    m->W = n_shards;
    m->use_rccl = use_rccl;
    m->T_total = T;
    m->nA = cfg->nNeutronA + cfg->nProtonA;
    m->nB = cfg->nNeutronB + cfg->nProtonB;
    m->nN = m->nA + m->nB;
    m->shards.resize(n_shards);
    for (int r = 0; r < n_shards; ++r) {
        auto &s = m->shards[r];
        s.device = devices[r];
        partition(T, r, n_shards, &s.offset, &s.count);
    }
    rc = for_each_shard(n_shards, [&](int r) {
        auto &s = m->shards[r];
        mcmc_config c = *cfg;
        c.device = s.device;
        c.event_offset = cfg->event_offset + s.offset;
        int e = create_handle(&c, s.count, &s.h);
        if (e) return e;
        s.fixed_ws_bytes = mcmc_row_sum_fixed_workspace_bytes((uint32_t)m->nN);
        MCMC_HIP_CHECK(hipMalloc(&s.d_fixed_ws, s.fixed_ws_bytes));
        return MCMC_OK;
    });
    if (rc == MCMC_OK && use_rccl) {
        std::vector<ncclComm_t> comms(n_shards);
        const ncclResult_t e = rccl().CommInitAll(comms.data(), n_shards, devices);
        if (e != ncclSuccess) rc = set_rccl_error(e, "ncclCommInitAll");
        else
            for (int r = 0; r < n_shards; ++r) m->shards[r].comm = comms[r];
    }
    if (rc != MCMC_OK) {
        const std::string keep = mcmc_last_error();
        (void)mcmc_multi_destroy(m);
        return set_error(rc, keep.c_str());
    }
    *out = m;
    return MCMC_OK;
}

extern "C" int mcmc_multi_destroy(mcmc_multi m)
{
    if (!m) return set_error(MCMC_EINVAL, "null handle");
    DeviceGuard restore_device;
    for (auto &s : m->shards) {
        (void)hipSetDevice(s.device);
        if (s.h) (void)hipStreamSynchronize(s.h->stream);
        if (s.comm) (void)rccl().CommDestroy(s.comm);
        if (s.d_vec) (void)hipFree(s.d_vec);
        if (s.d_fixed_ws) (void)hipFree(s.d_fixed_ws);
        if (s.h) (void)mcmc_destroy(s.h);
    }
    delete m;
    return MCMC_OK;
}

extern "C" int mcmc_multi_info(mcmc_multi m, int *n_shards, int *reduce_used, uint64_t *offsets, uint64_t *counts)
{
    if (!m) return set_error(MCMC_EINVAL, "null handle");
    if (n_shards) *n_shards = m->W;
    if (reduce_used) *reduce_used = m->use_rccl ? MCMC_REDUCE_RCCL : MCMC_REDUCE_HOST;
    for (int r = 0; r < m->W; ++r) {
        if (offsets) offsets[r] = m->shards[r].offset;
        if (counts) counts[r] = m->shards[r].count;
    }
    return MCMC_OK;
}

extern "C" mcmc_handle mcmc_multi_shard(mcmc_multi m, int shard)
{
    if (!m || shard < 0 || shard >= m->W) {
        (void)set_error(MCMC_EINVAL, "bad shard index");
        return nullptr;
    }
    return m->shards[shard].h;
}

extern "C" int mcmc_multi_warmup(mcmc_multi m, float stepsize, int iterations) // main.cu:149-152 on every shard
{
    if (!m) return set_error(MCMC_EINVAL, "null handle");
    DeviceGuard restore_device;
    return for_each_shard(m->W, [&](int r) { return mcmc_warmup(m->shards[r].h, stepsize, iterations); });
}

extern "C" int mcmc_multi_init_from_pdf(mcmc_multi m)
{
    if (!m) return set_error(MCMC_EINVAL, "null handle");
    DeviceGuard restore_device;
    return for_each_shard(m->W, [&](int r) { return mcmc_init_from_pdf(m->shards[r].h); });
}

// main.cu:157-286 over all shards: payload [totalRuns][nNucleons][4] (fixed-point limbs of the sum over the shard's
// threads of the per-thread r^2 means + the count of non-finite entries); mean over ALL threads = total / T
// (main.cu:221-223), then the run estimator of main.cu:253-258.
extern "C" int mcmc_multi_run_rms(mcmc_multi m, float *rms, float *rms_stderr)
{
    int rc = check_multi_scan(m);
    if (rc) return rc;
    if (!rms || !rms_stderr) return set_error(MCMC_EINVAL, "null pointer");
    DeviceGuard restore_device;
    const int R = m->cfg.totalRuns, nN = m->nN;
    const size_t len = (size_t)R * nN * 4;
    rc = for_each_shard(m->W, [&](int r) {
        auto &s = m->shards[r];
        mcmc_handle h = s.h;
        MCMC_HIP_CHECK(hipSetDevice(s.device));
        int e = ensure_vec(s, len);
        if (e) return e;
        for (int run = 0; run < R; ++run) {
            mcmc_xs_args a = xs_args(h, h->cfg.stepsize);
            if ((e = mcmc_launch_integrate_r2(&a, h->d_result, (uint32_t)h->cfg.samplesPerThread, nullptr, h->stream))) return e;
            h->sweeps_done += (uint64_t)h->cfg.samplesPerThread;
            if ((e = mcmc_launch_row_sum_fixed_f32(h->d_result, (uint32_t)nN, h->T, s.d_vec + (size_t)run * nN * 4, s.d_fixed_ws,
                                                   s.fixed_ws_bytes, h->stream)))
                return e;
        }
        return MCMC_OK;
    });
    if (rc) return rc;
    std::vector<double> tot;
    if ((rc = reduce_multi(m, len, tot)) != MCMC_OK) return rc;
    std::vector<float> ra(R);
    for (int n = 0; n < nN; ++n) {
        for (int run = 0; run < R; ++run) {
            const double *q = &tot[((size_t)run * nN + n) * 4];
            if (q[3] != 0.0) return set_error(MCMC_EINVAL, "mcmc_multi_run_rms: a per-thread r^2 mean was not finite");
            ra[run] = (float)(mcmc_fixed_sum_value(q) / (double)m->T_total); // main.cu:221-223
        }
        const float finalR2Average = mcmc_Average(ra.data(), R);                                  // main.cu:254
        const float finalR2Stderr = std::sqrt(mcmc_Variance(ra.data(), finalR2Average, R) / R);  // main.cu:255
        rms[n] = std::sqrt(finalR2Average);                                                      // main.cu:256
        rms_stderr[n] = finalR2Stderr / (2 * rms[n]);                                            // main.cu:257
    }
    return MCMC_OK;
}

// main.cu:289-413 over all shards: payload [ndbs][totalRuns][8] = {3 limbs of the sum over the shard's threads of the
// per-thread S_AB means, non-finite count, colliding pairs, samples with >= 1 collision, participants, samples}.
extern "C" int mcmc_multi_run_sab_scan(mcmc_multi m, float *b_out, float *S_AB, float *S_AB_stderr, double *p_inel, float dmax2)
{
    int rc = check_multi_scan(m);
    if (rc) return rc;
    if (!b_out || !S_AB || !S_AB_stderr) return set_error(MCMC_EINVAL, "null pointer");
    DeviceGuard restore_device;
    const int R = m->cfg.totalRuns, nb = m->cfg.ndbs;
    if (nb <= 0) return set_error(MCMC_EINVAL, "ndbs must be > 0");
    const size_t len = (size_t)nb * R * 8;
    rc = for_each_shard(m->W, [&](int r) {
        auto &s = m->shards[r];
        mcmc_handle h = s.h;
        MCMC_HIP_CHECK(hipSetDevice(s.device));
        int e = ensure_vec(s, len);
        if (e) return e;
        if (!p_inel) MCMC_HIP_CHECK(hipMemsetAsync(s.d_vec, 0, sizeof(double) * len, h->stream));
        for (int dbs = 0; dbs < nb; ++dbs) {
            const float b = h->cfg.b0 + dbs * h->cfg.db; // main.cu:328
            for (int run = 0; run < R; ++run) {
                double *slot = s.d_vec + ((size_t)dbs * R + run) * 8;
                mcmc_xs_args a = xs_args(h, h->cfg.stepsize);
                if ((e = mcmc_launch_integrate_sab(&a, h->d_result, (uint32_t)h->cfg.samplesPerThread, b, nullptr,
                                                   p_inel ? slot + 4 : nullptr, dmax2, h->stream)))
                    return e;
                h->sweeps_done += (uint64_t)h->cfg.samplesPerThread;
                if ((e = mcmc_launch_row_sum_fixed_f32(h->d_result, 1, h->T, slot, s.d_fixed_ws, s.fixed_ws_bytes, h->stream))) return e;
            }
        }
        return MCMC_OK;
    });
    if (rc) return rc;
    std::vector<double> tot;
    if ((rc = reduce_multi(m, len, tot)) != MCMC_OK) return rc;
    std::vector<float> runAverages(R);
    for (int dbs = 0; dbs < nb; ++dbs) {
        double inel = 0, nsamp = 0;
        for (int run = 0; run < R; ++run) {
            const double *q = &tot[((size_t)dbs * R + run) * 8];
            if (q[3] != 0.0) return set_error(MCMC_EINVAL, "mcmc_multi_run_sab_scan: a per-thread S_AB mean was not finite");
            runAverages[run] = (float)(mcmc_fixed_sum_value(q) / (double)m->T_total); // main.cu:359
            inel += q[5];
            nsamp += q[7];
        }
        const float finalAverage = mcmc_Average(runAverages.data(), R);                  // main.cu:384
        const float finalVariance = mcmc_Variance(runAverages.data(), finalAverage, R);  // main.cu:385
        b_out[dbs] = m->cfg.b0 + dbs * m->cfg.db;
        S_AB[dbs] = finalAverage;
        S_AB_stderr[dbs] = std::sqrt(finalVariance / R);                                 // main.cu:386
        if (p_inel) p_inel[dbs] = nsamp > 0 ? inel / nsamp : 0.0;
    }
    return MCMC_OK;
}

// the inelastic-flag scan (mcmc_run_inel_scan) over all shards: payload [ndbs][totalRuns][2] = {samples with a collision, samples}
extern "C" int mcmc_multi_run_inel_scan(mcmc_multi m, float *b_out, double *p_inel, double *p_inel_stderr, float dmax2)
{
    int rc = check_multi_scan(m);
    if (rc) return rc;
    if (!b_out || !p_inel) return set_error(MCMC_EINVAL, "null pointer");
    DeviceGuard restore_device;
    const int R = m->cfg.totalRuns, nb = m->cfg.ndbs;
    if (nb <= 0) return set_error(MCMC_EINVAL, "ndbs must be > 0");
    const size_t len = (size_t)nb * R * 2;
    rc = for_each_shard(m->W, [&](int r) {
        auto &s = m->shards[r];
        mcmc_handle h = s.h;
        MCMC_HIP_CHECK(hipSetDevice(s.device));
        int e = ensure_vec(s, len);
        if (e) return e;
        for (int dbs = 0; dbs < nb; ++dbs) {
            const float b = h->cfg.b0 + dbs * h->cfg.db;
            for (int run = 0; run < R; ++run) {
                mcmc_xs_args a = xs_args(h, h->cfg.stepsize);
                if ((e = mcmc_launch_integrate_inel(&a, (uint32_t)h->cfg.samplesPerThread, b, s.d_vec + ((size_t)dbs * R + run) * 2, dmax2,
                                                    h->stream)))
                    return e;
                h->sweeps_done += (uint64_t)h->cfg.samplesPerThread;
            }
        }
        return MCMC_OK;
    });
    if (rc) return rc;
    std::vector<double> tot;
    if ((rc = reduce_multi(m, len, tot)) != MCMC_OK) return rc;
    std::vector<double> pr(R);
    for (int dbs = 0; dbs < nb; ++dbs) {
        for (int run = 0; run < R; ++run) {
            const double *q = &tot[((size_t)dbs * R + run) * 2];
            pr[run] = q[1] > 0 ? q[0] / q[1] : 0.0;
        }
        b_out[dbs] = m->cfg.b0 + dbs * m->cfg.db;
        p_inel[dbs] = mcmc_Average_f64(pr.data(), R);
        if (p_inel_stderr) p_inel_stderr[dbs] = R > 1 ? std::sqrt(mcmc_Variance_f64(pr.data(), R) / R) : 0.0;
    }
    return MCMC_OK;
}

// =====================================================================
// stepsize (stepsize/src/main.cpp:17-30 for many chains) on one or several GPUs.
namespace {

// FP32 launches keep their lag products in FP32 per chain (include/mcmc_hip.h, mcmc_chain_run): a window longer than
// this is run as ring-carried launches whose FP64 statistics the host adds.
constexpr uint64_t STEPSIZE_SEGMENT = 65536;

// Mean of the target, by quadrature (midpoint rule on 8192 bins): the accumulators are centred on it so that the FP32
// sums of y^2 and y_i y_{i+k} see deviations, not positions (statistics are shift-invariant, statistics.hpp:17-57).
// Deterministic and the same on every shard (a warm-up mean would need an exchange of its own).
double target_mean(const mcmc_target &t, double x0)
{
    // every density of Distributions.cuh:7-28 / metropolis.hpp:9-17 is 0 for r < 0: the integrals run over [0, hi]
    const int n = 8192;
    double hi, s0 = 0, s1 = 0;
    if (t.kind == MCMC_TARGET_WS || t.kind == MCMC_TARGET_R2WS) hi = (double)t.p0 + 40.0 * (double)t.p1;
    else hi = 12.0 * std::sqrt((double)t.p0);
    if (!(hi > 0.0) || !std::isfinite(hi)) return x0;
    const double h = hi / n;
    for (int i = 0; i < n; ++i) {
        const double r = (i + 0.5) * h;
        double p;
        switch (t.kind) {
        case MCMC_TARGET_WS: p = 1 / (1 + std::exp((r - (double)t.p0) / (double)t.p1)); break;         // Distributions.cuh:7-10
        case MCMC_TARGET_R2WS: p = r * r / (1 + std::exp((r - (double)t.p0) / (double)t.p1)); break;   // metropolis.hpp:9-12
        case MCMC_TARGET_GAUSS: p = std::exp(-r * r / (2 * (double)t.p0)); break;                      // Distributions.cuh:19-22
        default: p = r * r * std::exp(-r * r / (2 * (double)t.p0)); break;                             // metropolis.hpp:14-17
        }
        s0 += p;
        s1 += p * r;
    }
    return s0 > 0.0 && std::isfinite(s1 / s0) ? s1 / s0 : x0;
}

struct StepsizeShard {
    int device = 0;
    uint64_t offset = 0, count = 0;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    void *d_state = nullptr, *d_ws = nullptr, *d_samples = nullptr, *d_ring = nullptr;
    double *d_stats = nullptr;
    size_t state_bytes = 0, ws_bytes = 0, samples_bytes = 0, ring_bytes = 0, stats_bytes = 0;
    std::vector<char> init;

    static hipError_t grow(void **p, size_t *have, size_t need)
    {
        if (*have >= need) return hipSuccess;
        if (*p) (void)hipFree(*p);
        *p = nullptr;
        *have = 0;
        const hipError_t e = hipMalloc(p, need);
        if (e == hipSuccess) *have = need;
        return e;
    }
    void release()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        if (comm) (void)rccl().CommDestroy(comm);
        for (void *p : {d_state, d_ws, d_samples, d_ring, (void *)d_stats})
            if (p) (void)hipFree(p);
        if (stream) (void)hipStreamDestroy(stream);
        d_state = d_ws = d_samples = d_ring = nullptr;
        d_stats = nullptr;
        stream = nullptr;
        comm = nullptr;
    }
};

// The shards and their buffers outlive one width (mcmc_tune_stepsize runs a grid of widths on them).
struct StepsizeJob {
    std::vector<StepsizeShard> shards;
    bool use_rccl = false;
    ~StepsizeJob()
    {
        for (auto &s : shards) s.release();
    }

    int open(const int *devices, int n_shards, int reduce)
    {
        int rc = check_devices(devices, n_shards);
        if (rc) return rc;
        if ((rc = resolve_reduce(reduce, devices, n_shards, &use_rccl)) != MCMC_OK) return rc;
        shards.resize(n_shards);
        for (int r = 0; r < n_shards; ++r) {
            shards[r].device = devices[r];
            MCMC_HIP_CHECK(hipSetDevice(devices[r]));
            MCMC_HIP_CHECK(hipStreamCreate(&shards[r].stream));
        }
        if (use_rccl) {
            std::vector<ncclComm_t> comms(n_shards);
            const ncclResult_t e = rccl().CommInitAll(comms.data(), n_shards, devices);
            if (e != ncclSuccess) return set_rccl_error(e, "ncclCommInitAll");
            for (int r = 0; r < n_shards; ++r) shards[r].comm = comms[r];
        }
        return MCMC_OK;
    }

    int run(const mcmc_target *target, int precision, int accept_rule, int math_mode, double step, uint64_t n_chains,
            uint32_t n_coords, uint64_t n_warmup, uint64_t n_steps, uint32_t n_lags, uint64_t seed, double x0,
            mcmc_chain_summary *summary, double *rho, void *samples_host)
    {
        if (!target || !summary) return set_error(MCMC_EINVAL, "null pointer");
        if (n_chains == 0 || n_coords == 0 || n_steps == 0) return set_error(MCMC_EINVAL, "n_chains, n_coords and n_steps must be > 0");
        if (precision != 32 && precision != 64) return set_error(MCMC_EINVAL, "precision must be 32 or 64");
        if (n_lags > MCMC_MAX_LAGS) return set_error(MCMC_EINVAL, "n_lags must be <= MCMC_MAX_LAGS (49)");
        const int W = (int)shards.size();
        if (n_chains < (uint64_t)W) return set_error(MCMC_EINVAL, "fewer chains than shards");
        if (samples_host && W != 1) return set_error(MCMC_EINVAL, "sample write-back to the host needs one shard");
        const size_t esz = precision == 64 ? 8 : 4;
        const int len = MCMC_STATS_LEN(n_lags);

        // The statistics window as launches: the FP64 kernel folds its lag products into FP64 as it goes; the FP32
        // ones are cut into segments carried by `ring` (the last segment absorbs a remainder shorter than n_lags).
        std::vector<uint64_t> seg;
        if (precision == 32 && n_lags > 0 && n_steps > STEPSIZE_SEGMENT) {
            uint64_t left = n_steps;
            while (left > 0) {
                uint64_t n = left < STEPSIZE_SEGMENT ? left : STEPSIZE_SEGMENT;
                if (left - n > 0 && left - n < (uint64_t)n_lags + 1) n = left;
                seg.push_back(n);
                left -= n;
            }
        } else seg.push_back(n_steps);
        const size_t n_seg = seg.size();
        const size_t vec_len = n_seg * n_coords * (size_t)len;
        const double center = target_mean(*target, x0);

        for (int r = 0; r < W; ++r) partition(n_chains, r, W, &shards[r].offset, &shards[r].count);
        int rc = for_each_shard(W, [&](int r) {
            StepsizeShard &s = shards[r];
            MCMC_HIP_CHECK(hipSetDevice(s.device));
            const size_t nstate = (size_t)s.count * n_coords;
            mcmc_chain_args a{};
            a.n_chains = s.count;
            a.n_coords = n_coords;
            a.n_lags = n_lags;
            a.step_size = step;
            a.center = center;
            a.target = *target;
            a.precision = precision;
            a.accept_rule = accept_rule;
            a.math_mode = math_mode;
            a.seed = seed;
            a.chain_offset = s.offset;
            const size_t ws_bytes = mcmc_chain_workspace_bytes(&a);
            MCMC_HIP_CHECK(StepsizeShard::grow(&s.d_state, &s.state_bytes, nstate * esz));
            MCMC_HIP_CHECK(StepsizeShard::grow(&s.d_ws, &s.ws_bytes, ws_bytes));
            MCMC_HIP_CHECK(StepsizeShard::grow((void **)&s.d_stats, &s.stats_bytes, sizeof(double) * vec_len));
            if (samples_host) MCMC_HIP_CHECK(StepsizeShard::grow(&s.d_samples, &s.samples_bytes, (size_t)n_steps * nstate * esz));
            if (n_seg > 1) MCMC_HIP_CHECK(StepsizeShard::grow(&s.d_ring, &s.ring_bytes, (size_t)n_lags * nstate * esz));
            a.workspace = s.d_ws;
            a.workspace_bytes = ws_bytes;
            // x0 (metropolis.hpp:29 `double r = 1`)
            s.init.resize(nstate * esz);
            for (size_t i = 0; i < nstate; i++) {
                if (precision == 64) ((double *)s.init.data())[i] = x0;
                else ((float *)s.init.data())[i] = (float)x0;
            }
            MCMC_HIP_CHECK(hipMemcpyAsync(s.d_state, s.init.data(), s.init.size(), hipMemcpyHostToDevice, s.stream));
            a.state = s.d_state;
            a.stats = s.d_stats;
            int e;
            if (n_warmup > 0) {
                a.n_lags = 0;
                a.n_steps = n_warmup;
                a.step_offset = 0;
                a.window_start = 1;
                if ((e = mcmc_chain_run(&a, s.stream)) != MCMC_OK) return e;
            }
            a.n_lags = n_lags;
            a.ring = n_seg > 1 ? s.d_ring : nullptr;
            uint64_t done = 0;
            for (size_t j = 0; j < n_seg; ++j) {
                a.n_steps = seg[j];
                a.step_offset = n_warmup + done;
                a.window_start = j == 0;
                a.stats = s.d_stats + j * n_coords * (size_t)len;
                a.samples = samples_host ? (char *)s.d_samples + done * nstate * esz : nullptr;
                if ((e = mcmc_chain_run(&a, s.stream)) != MCMC_OK) return e;
                done += seg[j];
            }
            return MCMC_OK;
        });
        if (rc) return rc;
        std::vector<ReducePeer> peers(W);
        for (int r = 0; r < W; ++r) peers[r] = ReducePeer{shards[r].device, shards[r].stream, shards[r].d_stats, shards[r].comm};
        std::vector<double> tot;
        if ((rc = all_reduce_sum(peers, vec_len, use_rccl, tot)) != MCMC_OK) return rc;
        if (samples_host) {
            StepsizeShard &s = shards[0];
            MCMC_HIP_CHECK(hipSetDevice(s.device));
            MCMC_HIP_CHECK(hipMemcpyAsync(samples_host, s.d_samples, (size_t)n_steps * s.count * n_coords * esz, hipMemcpyDeviceToHost, s.stream));
            MCMC_HIP_CHECK(hipStreamSynchronize(s.stream));
        }
        // window statistics = sum over segments of [0, 3 + L); head sums A from the first segment, tail sums Z from
        // the last (include/mcmc_hip.h, mcmc_stats_finalize)
        std::vector<double> win(len);
        for (uint32_t c = 0; c < n_coords; c++) {
            std::fill(win.begin(), win.end(), 0.0);
            for (size_t j = 0; j < n_seg; ++j) {
                const double *st = &tot[(j * n_coords + c) * (size_t)len];
                for (uint32_t i = 0; i < 3 + n_lags; ++i) win[i] += st[i];
                if (j == 0)
                    for (uint32_t i = 0; i < n_lags; ++i) win[3 + n_lags + i] = st[3 + n_lags + i];
                if (j + 1 == n_seg)
                    for (uint32_t i = 0; i < n_lags; ++i) win[3 + 2 * n_lags + i] = st[3 + 2 * n_lags + i];
            }
            rc = mcmc_stats_finalize(win.data(), n_lags, n_chains, n_steps, center, &summary[c], rho ? rho + (size_t)c * n_lags : nullptr);
            if (rc) return rc;
        }
        return MCMC_OK;
    }
};

int current_device(int *dev)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_error(MCMC_ENODEV, "no HIP device");
    MCMC_HIP_CHECK(hipGetDevice(dev));
    return MCMC_OK;
}

} // namespace

extern "C" int mcmc_stepsize_run_multi(const mcmc_target *target, int precision, int accept_rule, int math_mode, double step,
                                       uint64_t n_chains, uint32_t n_coords, uint64_t n_warmup, uint64_t n_steps, uint32_t n_lags,
                                       uint64_t seed, double x0, const int *devices, int n_shards, int reduce,
                                       mcmc_chain_summary *summary, double *rho, void *samples_host)
{
    DeviceGuard restore_device; // declared first: the job's destructor switches devices too
    StepsizeJob job;
    int rc = job.open(devices, n_shards, reduce);
    if (rc) return rc;
    return job.run(target, precision, accept_rule, math_mode, step, n_chains, n_coords, n_warmup, n_steps, n_lags, seed, x0, summary,
                   rho, samples_host);
}

extern "C" int mcmc_stepsize_run(const mcmc_target *target, int precision, int accept_rule, int math_mode, double step,
                                 uint64_t n_chains, uint32_t n_coords, uint64_t n_warmup, uint64_t n_steps, uint32_t n_lags,
                                 uint64_t seed, double x0, mcmc_chain_summary *summary, double *rho, void *samples_host)
{
    int dev = 0;
    int rc = current_device(&dev);
    if (rc) return rc;
    return mcmc_stepsize_run_multi(target, precision, accept_rule, math_mode, step, n_chains, n_coords, n_warmup, n_steps, n_lags,
                                   seed, x0, &dev, 1, MCMC_REDUCE_HOST, summary, rho, samples_host);
}

// The step-size study of the stepsize executable (stepsize/src/main.cpp:14-36: acceptance and lag-k
// autocorrelation per proposal width) run on the device for a grid of widths, and the width of least
// integrated autocorrelation time 1 + 2 sum_k rho_k (the notebook's efficiency, StepSizeAnalysis.ipynb
// cell 1), refined by a parabola through the grid minimum and its neighbours.  Closes the loop to
// xsection, whose width is the constant 2.2f of main.cu:149,197,335.
extern "C" int mcmc_tune_stepsize(const mcmc_target *target, double lo, double hi, int n_widths, uint64_t n_chains,
                                  uint64_t n_warmup, uint64_t n_steps, uint64_t seed, double x0, double *widths,
                                  double *acceptance, double *tau_int, double *best_width)
{
    if (!target || !best_width) return set_error(MCMC_EINVAL, "null pointer");
    if (!(lo > 0.0) || !(hi > lo) || n_widths < 3) return set_error(MCMC_EINVAL, "need 0 < lo < hi and n_widths >= 3");
    if (n_steps <= (uint64_t)MCMC_MAX_LAGS + 1) return set_error(MCMC_EINVAL, "n_steps must exceed MCMC_MAX_LAGS + 1");
    std::vector<double> w(n_widths), acc(n_widths), tau(n_widths);
    DeviceGuard restore_device;
    int dev = 0;
    int rc = current_device(&dev);
    if (rc) return rc;
    StepsizeJob job; // one allocation for all widths
    if ((rc = job.open(&dev, 1, MCMC_REDUCE_HOST)) != MCMC_OK) return rc;
    for (int i = 0; i < n_widths; ++i) {
        w[i] = lo + (hi - lo) * i / (n_widths - 1);
        mcmc_chain_summary s;
        rc = job.run(target, 32, MCMC_ACCEPT_CROSS, MCMC_MATH_FAST, w[i], n_chains, 1, n_warmup, n_steps, MCMC_MAX_LAGS, seed, x0, &s,
                     nullptr, nullptr);
        if (rc) return rc;
        acc[i] = s.acceptance;
        tau[i] = s.tau_int;
    }
    int k = 0;
    for (int i = 1; i < n_widths; ++i)
        if (tau[i] < tau[k]) k = i;
    double best = w[k];
    if (k > 0 && k + 1 < n_widths) { // vertex of the parabola through (w[k-1..k+1], tau[k-1..k+1]); equal spacing
        const double d = tau[k - 1] - 2.0 * tau[k] + tau[k + 1];
        if (d > 0.0) best = w[k] + 0.5 * (w[k] - w[k - 1]) * (tau[k - 1] - tau[k + 1]) / d;
    }
    for (int i = 0; i < n_widths; ++i) {
        if (widths) widths[i] = w[i];
        if (acceptance) acceptance[i] = acc[i];
        if (tau_int) tau_int[i] = tau[i];
    }
    *best_width = best;
    return MCMC_OK;
}
