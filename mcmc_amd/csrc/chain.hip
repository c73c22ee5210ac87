// chain.hip -- host side of the many-chain MH kernel: argument checks, kernel
// selection, launch; and the host finalisation of the statistics vector with
// the exact formulas of stepsize/include/statistics.hpp:6-57.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "chain_lag_kernel.hpp"
#include "host_common.hpp"

using namespace mcmc;

namespace mcmc {
// one translation unit per target kind (chain_kind.hip)
int chain_dispatch_kind_0(const mcmc_chain_args *, const ChainParams &, dim3, hipStream_t);
int chain_dispatch_kind_1(const mcmc_chain_args *, const ChainParams &, dim3, hipStream_t);
int chain_dispatch_kind_2(const mcmc_chain_args *, const ChainParams &, dim3, hipStream_t);
int chain_dispatch_kind_3(const mcmc_chain_args *, const ChainParams &, dim3, hipStream_t);
} // namespace mcmc

namespace mcmc {
int g_lag_mfma = -1; // MCMC_LAG_MFMA: 0 = lag products on the VALU (chain_kernel), otherwise chain_lag_kernel; -1 = not read yet
int g_samples_pairs = -1; // MCMC_SAMPLES_PAIRS: 1 = sample write-back in the pair layout; -1 = not read yet
}

namespace {

bool lag_mfma_enabled()
{
    if (g_lag_mfma < 0) {
        const char *v = std::getenv("MCMC_LAG_MFMA");
        g_lag_mfma = v && v[0] == '0' ? 0 : 1;
    }
    return g_lag_mfma != 0;
}

bool samples_pairs_enabled()
{
    if (g_samples_pairs < 0) {
        const char *v = std::getenv("MCMC_SAMPLES_PAIRS");
        g_samples_pairs = v && v[0] == '1' ? 1 : 0;
    }
    return g_samples_pairs != 0;
}

// chain_lag_kernel stores y s as two f16 halves: s = the power of two that maps the farthest |x - center| a chain of this
// target can reach (40 diffuseness lengths / 12 standard deviations beyond the bulk, plus 8 proposal widths) to <= 2^15.
// Samples beyond it (a chain started far away) saturate at 65000 / s.
float lag_scale_for(const mcmc_chain_args *a)
{
    const mcmc_target &t = a->target;
    double reach = t.kind == MCMC_TARGET_WS || t.kind == MCMC_TARGET_R2WS ? std::fabs((double)t.p0) + 40.0 * std::fabs((double)t.p1)
                                                                             : 12.0 * std::sqrt(std::fabs((double)t.p0));
    reach += std::fabs(a->center) + 8.0 * std::fabs(a->step_size);
    if (!(reach > 0.0) || !std::isfinite(reach)) reach = 1.0;
    int e = 15 - (int)std::ceil(std::log2(reach));
    if (e > 100) e = 100;
    if (e < -100) e = -100;
    return std::ldexp(1.0f, e);
}

uint32_t chain_grid_x(uint64_t n_chains)
{
    const uint64_t tiles = (n_chains + CHAIN_BLOCK - 1) / CHAIN_BLOCK;
    // 256 CUs x 8 resident 256-thread blocks; more tiles than that are looped
    // over inside the kernel so that the partials buffer stays small.
    return (uint32_t)(tiles < 2048 ? tiles : 2048);
}

} // namespace

extern "C" size_t mcmc_chain_workspace_bytes(const mcmc_chain_args *a)
{
    if (!a) return 0;
    return (size_t)a->n_coords * chain_grid_x(a->n_chains) * MCMC_STATS_LEN(a->n_lags) * sizeof(double);
}

namespace {
// Argument checks + both launches of mcmc_chain_run.  `workspace` holds the block partials (the caller's buffer, or
// the library's static scratch for mcmc_chain_run_noalloc); `stamps` != nullptr selects the stamped instantiation.
int chain_launch(const mcmc_chain_args *a, void *workspace, size_t workspace_bytes, uint64_t *stamps, void *stream)
{
    if (!a) return set_error(MCMC_EINVAL, "null args");
    if (a->n_chains == 0 || a->n_coords == 0) return set_error(MCMC_EINVAL, "n_chains and n_coords must be > 0");
    if (a->n_coords > 65535) return set_error(MCMC_EINVAL, "n_coords must be <= 65535");
    if (a->n_chains > (1ull << 39)) return set_error(MCMC_EINVAL, "n_chains must be <= 2^39");
    // the Philox counter carries 40 bits of the global chain id (device_common.hpp draw_block)
    if (a->chain_offset > (1ull << 40) || a->chain_offset + a->n_chains > (1ull << 40))
        return set_error(MCMC_EINVAL, "chain_offset + n_chains must be <= 2^40 (global chain ids are 40-bit Philox counters)");
    if (a->n_lags > MCMC_MAX_LAGS) return set_error(MCMC_EINVAL, "n_lags must be <= MCMC_MAX_LAGS (49)");
    if (!a->state || !a->stats) return set_error(MCMC_EINVAL, "state and stats must be device pointers");
    if ((a->inj_normals == nullptr) != (a->inj_uniforms == nullptr))
        return set_error(MCMC_EINVAL, "inj_normals and inj_uniforms must be given together");
    if (a->n_lags > 0 && a->window_start && a->n_steps < a->n_lags)
        return set_error(MCMC_EINVAL, "the first launch of a statistics window needs n_steps >= n_lags");
    const size_t need = mcmc_chain_workspace_bytes(a);
    if (!workspace || workspace_bytes < need) return set_error(MCMC_ENOMEM, "workspace smaller than mcmc_chain_workspace_bytes()");
    if (a->target.kind < 0 || a->target.kind > 3) return set_error(MCMC_EINVAL, "bad target kind");

    ChainParams p;
    p.n_chains = a->n_chains;
    p.n_steps = a->n_steps;
    p.seed = a->seed;
    p.step_offset = a->step_offset;
    p.chain_offset = a->chain_offset;
    p.n_coords = a->n_coords;
    p.n_lags = a->n_lags;
    p.n_tiles = (uint32_t)((a->n_chains + CHAIN_BLOCK - 1) / CHAIN_BLOCK);
    p.window_start = a->window_start;
    p.step = a->step_size;
    p.center = a->center;
    p.target = a->target;
    p.state = a->state;
    p.partials = (double *)workspace;
    p.ring = a->ring;
    p.samples = a->samples;
    p.inj_normals = a->inj_normals;
    p.inj_uniforms = a->inj_uniforms;
    p.accept_bits = a->accept_bits;
    p.stamps = stamps;
    p.lag_mfma = a->precision == 32 && a->n_lags > 0 && lag_mfma_enabled();
    p.lag_scale = lag_scale_for(a);
    p.samples_pairs = 0;
    if (a->samples != nullptr && samples_pairs_enabled()) {
        // the pair layout exists for the throughput form of the write-back path only (mcmc_hip.h)
        if (a->precision != 32 || a->math_mode != MCMC_MATH_FAST || a->n_lags != 0 || a->inj_normals || a->accept_bits ||
            a->n_chains >= (1ull << 29) || (a->step_offset & 1) || (a->n_steps & 1))
            return set_error(MCMC_EINVAL, "MCMC_SAMPLES_PAIRS: FP32 FAST-math Philox launches without lags or accept flags, "
                                          "step_offset and n_steps even, n_chains < 2^29");
        p.samples_pairs = 1;
    }

    hipStream_t s = (hipStream_t)stream;
    // FP32 with lags: chain_lag_kernel, one 256-chain tile per workgroup like chain_kernel (the workspace is sized for either)
    const dim3 grid(p.lag_mfma ? lag_grid_x(a->n_chains) : chain_grid_x(a->n_chains), a->n_coords);
    int rc;
    switch (a->target.kind) {
    case MCMC_TARGET_WS: rc = chain_dispatch_kind_0(a, p, grid, s); break;
    case MCMC_TARGET_R2WS: rc = chain_dispatch_kind_1(a, p, grid, s); break;
    case MCMC_TARGET_GAUSS: rc = chain_dispatch_kind_2(a, p, grid, s); break;
    default: rc = chain_dispatch_kind_3(a, p, grid, s); break;
    }
    if (rc != MCMC_OK) return rc;
    MCMC_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(chain_finalize_kernel, dim3(MCMC_STATS_LEN(a->n_lags), a->n_coords), dim3(256), 0, s,
                       (const double *)workspace, a->stats, (int)MCMC_STATS_LEN(a->n_lags), (int)grid.x);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}
} // namespace

// ---------------------------------------------------------------------
// mcmc_chain_run for callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc): the block partials go to a file-scope device array instead of a caller workspace.  Every GPU the
// library runs on holds its own copy of the array at its own address; the address is looked up once per device
// (first call on that device, with it current) and only that pointer is ever handed to kernels.
namespace {
constexpr size_t CHAIN_SCRATCH_BYTES = MCMC_CHAIN_NOALLOC_MAX_BYTES;
static __device__ __align__(256) unsigned char chain_scratch[CHAIN_SCRATCH_BYTES];
constexpr int SCRATCH_MAX_DEVICES = 64;
std::mutex scratch_mutex;
std::atomic<void *> scratch_ptr[SCRATCH_MAX_DEVICES]; // per device; nullptr = not looked up yet

int chain_scratch_for_current_device(void **out)
{
    int dev = -1;
    MCMC_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= SCRATCH_MAX_DEVICES) return set_error(MCMC_EINVAL, "device ordinal beyond the scratch table");
    void *ptr = scratch_ptr[dev].load(std::memory_order_acquire);
    if (ptr == nullptr) {
        std::lock_guard<std::mutex> lock(scratch_mutex);
        ptr = scratch_ptr[dev].load(std::memory_order_relaxed);
        if (ptr == nullptr) {
            MCMC_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(chain_scratch)));
            if (ptr == nullptr) return set_error(MCMC_EINVAL, "hipGetSymbolAddress returned a null scratch address");
            scratch_ptr[dev].store(ptr, std::memory_order_release);
        }
    }
    *out = ptr;
    return MCMC_OK;
}
} // namespace

extern "C" int mcmc_chain_run(const mcmc_chain_args *a, void *stream)
{
    if (!a) return set_error(MCMC_EINVAL, "null args");
    // Resolve the no-allocation entry point's scratch address for this device now (once per device; afterwards one relaxed
    // load): a caller that has run ANY chain launch on a device can enter a capture / no-allocation region and call
    // mcmc_chain_run_noalloc there without that call doing anything but launching.  Failure here is not this call's error.
    void *unused = nullptr;
    (void)chain_scratch_for_current_device(&unused);
    return chain_launch(a, a->workspace, a->workspace_bytes, nullptr, stream);
}

extern "C" int mcmc_chain_run_noalloc(const mcmc_chain_args *a, void *stream)
{
    if (!a) return set_error(MCMC_EINVAL, "null args");
    const size_t need = mcmc_chain_workspace_bytes(a);
    if (need > CHAIN_SCRATCH_BYTES)
        return set_error(MCMC_ENOMEM, "mcmc_chain_run_noalloc: n_coords * blocks * MCMC_STATS_LEN(n_lags) * 8 bytes exceeds "
                                      "MCMC_CHAIN_NOALLOC_MAX_BYTES; use mcmc_chain_run with a workspace");
    void *ptr = nullptr;
    const int rc = chain_scratch_for_current_device(&ptr);
    if (rc != MCMC_OK) return rc;
    // every partial the finalisation reads is written by this launch's blocks first: nothing to clear
    return chain_launch(a, ptr, CHAIN_SCRATCH_BYTES, nullptr, stream);
}

// Diagnostic (no reference counterpart): the shader clock the chain kernel sustains.  Launches the STAMPED twin of the
// kernel these arguments select (same body + one s_memtime / s_memrealtime pair per block around its tile loop), and
// returns the median over blocks of shader cycles per 100 MHz tick, in GHz (MI355X_MICROARCH.md "DVFS give-back" (6)).
// Advances the chains like mcmc_chain_run.  Synchronous; allocates its own stamp buffer.
extern "C" int mcmc_chain_clock_probe(const mcmc_chain_args *a, double *clock_ghz, void *stream)
{
    if (!a || !clock_ghz) return set_error(MCMC_EINVAL, "null pointer");
    if (a->n_chains == 0 || a->n_coords == 0) return set_error(MCMC_EINVAL, "n_chains and n_coords must be > 0");
    const size_t nblk = (size_t)chain_grid_x(a->n_chains) * a->n_coords;
    uint64_t *d_stamps = nullptr;
    MCMC_HIP_CHECK(hipMalloc((void **)&d_stamps, nblk * 2 * sizeof(uint64_t)));
    int rc = MCMC_OK;
    hipError_t e = hipMemsetAsync(d_stamps, 0, nblk * 2 * sizeof(uint64_t), (hipStream_t)stream);
    std::vector<uint64_t> h(nblk * 2);
    if (e == hipSuccess) rc = chain_launch(a, a->workspace, a->workspace_bytes, d_stamps, stream);
    if (e == hipSuccess && rc == MCMC_OK)
        e = hipMemcpyAsync(h.data(), d_stamps, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess && rc == MCMC_OK) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d_stamps);
    if (rc != MCMC_OK) return rc;
    if (e != hipSuccess) return set_hip_error(e, __FILE__, __LINE__);
    if (const char *dump = std::getenv("MCMC_STAMP_DUMP")) { // development: the raw stamps, two 64-bit words per workgroup
        if (FILE *f = std::fopen(dump, "wb")) {
            std::fwrite(h.data(), sizeof(uint64_t), h.size(), f);
            std::fclose(f);
        }
    }
    std::vector<double> ratio;
    for (size_t b = 0; b < nblk; ++b)
        if (h[2 * b + 1] > 0) ratio.push_back((double)h[2 * b] / (double)h[2 * b + 1]);
    if (ratio.empty()) return set_error(MCMC_EINVAL, "the launch was too short for the 100 MHz counter to tick");
    std::nth_element(ratio.begin(), ratio.begin() + ratio.size() / 2, ratio.end());
    *clock_ghz = ratio[ratio.size() / 2] * 0.1;
    return MCMC_OK;
}

// statistics.hpp:6-57 evaluated from sufficient statistics.  With y = x - center:
//   Average      = S1/N + center                               (:6-15)
//   Variance     = sum (y-mu)^2 / (N-1) = (S2 - N mu^2)/(N-1)  (:17-28)
//   AutoCorr(k)  = [sum_{i<n-k} (y_i-mu)(y_{i+k}-mu) / (count_k - 1)] / Variance   (:36-45)
// where sum (y_i-mu)(y_{i+k}-mu) = P_k - mu (F_k + B_k) + count_k mu^2,
// F_k = S1 - sum_{j<k} Z_j (all but the last k samples), B_k = S1 - sum_{i<k} A_i.
extern "C" int mcmc_stats_finalize(const double *stats, uint32_t n_lags, uint64_t n_chains, uint64_t n_steps_total,
                                   double center, mcmc_chain_summary *out, double *rho)
{
    if (!stats || !out) return set_error(MCMC_EINVAL, "null pointer");
    if (n_chains == 0 || n_steps_total == 0) return set_error(MCMC_EINVAL, "empty window");
    const double N = (double)n_chains * (double)n_steps_total;
    const double S1 = stats[0], S2 = stats[1];
    const double mu = S1 / N;
    out->mean = mu + center;
    out->variance = (S2 - N * mu * mu) / (N - 1.0);
    out->acceptance = stats[2] / N;
    double tau = 1.0;
    double headsum = 0.0, tailsum = 0.0;
    for (uint32_t k = 1; k <= n_lags; ++k) {
        headsum += stats[3 + n_lags + (k - 1)];
        tailsum += stats[3 + 2 * n_lags + (k - 1)];
        double r = 0.0;
        if (n_steps_total > k) {
            const double cnt = (double)n_chains * (double)(n_steps_total - k);
            const double F = S1 - tailsum, B = S1 - headsum;
            const double cov = stats[3 + (k - 1)] - mu * (F + B) + cnt * mu * mu;
            r = cov / (cnt - 1.0) / out->variance;
        }
        if (rho) rho[k - 1] = r;
        tau += 2.0 * r;
    }
    out->tau_int = tau;
    return MCMC_OK;
}

// ---------------------------------------------------------------------
// The library's draws as arrays: what RandomNormal / RandomUniform
// (stepsize/include/random.hpp:7-19) and curand_normal / curand_uniform
// (Metropolis.cuh:43,47; IntegrateSAB.cuh:51-52) are replaced by.
namespace {
// normals / uniforms: the packed MH stream of the chain kernels (TAG_MH);
// u0s / u1s: the angle uniforms of the S_AB kernel (TAG_ANGLE);
// sab_normals / sab_uniforms: the MH stream of the S_AB kernel (TAG_SAB_MH).
template <int MATH>
__global__ void draws_kernel(uint64_t seed, uint64_t chain_offset, uint64_t n_chains, uint32_t coord, uint64_t step_offset,
                             uint64_t n_steps, float *normals, float *uniforms, float *u0s, float *u1s, float *sab_normals,
                             float *sab_uniforms)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chains) return;
    const uint64_t ta = step_offset, tb = step_offset + n_steps;
    if (n_steps == 0) return;
    const uint64_t id = chain_offset + c;
    if (normals || uniforms) {
        for (uint64_t g = ta >> 3; g <= ((tb - 1) >> 3); ++g) {
            Spare sp = {};
            u32x4 w[4];
            w[0] = pair_words<0>(seed, id, coord, g, sp);
            w[1] = pair_words<1>(seed, id, coord, g, sp);
            w[2] = pair_words<2>(seed, id, coord, g, sp);
            w[3] = pair_words<3>(seed, id, coord, g, sp);
            for (int j = 0; j < 4; ++j) {
                float z0, z1, v0, v1;
                pair_draws_f32<MATH, false>(w[j], 0.0f, z0, z1, v0, v1);
                const uint64_t t0 = (g << 3) + 2 * j;
                if (t0 >= ta && t0 < tb) {
                    if (normals) normals[(t0 - ta) * n_chains + c] = z0;
                    if (uniforms) uniforms[(t0 - ta) * n_chains + c] = v0;
                }
                if (t0 + 1 >= ta && t0 + 1 < tb) {
                    if (normals) normals[(t0 + 1 - ta) * n_chains + c] = z1;
                    if (uniforms) uniforms[(t0 + 1 - ta) * n_chains + c] = v1;
                }
            }
        }
    }
    if (!(u0s || u1s || sab_normals || sab_uniforms)) return;
    for (uint64_t blk = ta >> 1; blk <= ((tb - 1) >> 1); ++blk) {
        const u32x4 r = draw_block(seed, id, coord, blk, TAG_SAB_MH);
        const u32x4 q = draw_block(seed, id, coord, blk, TAG_ANGLE);
        float z0, z1;
        box_muller<MATH>(r.x, r.y, z0, z1);
        const uint64_t te = blk << 1;
        if (te >= ta) {
            const uint64_t o = (te - ta) * n_chains + c;
            if (sab_normals) sab_normals[o] = z0;
            if (sab_uniforms) sab_uniforms[o] = u01(r.z);
            if (u0s) u0s[o] = u01(q.x);
            if (u1s) u1s[o] = u01(q.y);
        }
        if (te + 1 < tb) {
            const uint64_t o = (te + 1 - ta) * n_chains + c;
            if (sab_normals) sab_normals[o] = z1;
            if (sab_uniforms) sab_uniforms[o] = u01(r.w);
            if (u0s) u0s[o] = u01(q.z);
            if (u1s) u1s[o] = u01(q.w);
        }
    }
}
} // namespace

extern "C" int mcmc_launch_draws(uint64_t seed, uint64_t chain_offset, uint64_t n_chains, uint32_t coord,
                                 uint64_t step_offset, uint64_t n_steps, int math_mode, float *normals, float *uniforms,
                                 float *angle_u0, float *angle_u1, float *sab_normals, float *sab_uniforms, void *stream)
{
    if (n_chains == 0) return set_error(MCMC_EINVAL, "n_chains must be > 0");
    if (n_chains > (1ull << 38)) return set_error(MCMC_EINVAL, "n_chains must be <= 2^38 per launch");
    if (chain_offset > (1ull << 40) || chain_offset + n_chains > (1ull << 40))
        return set_error(MCMC_EINVAL, "chain_offset + n_chains must be <= 2^40 (global chain ids are 40-bit Philox counters)");
    if (coord >= (1u << 24)) return set_error(MCMC_EINVAL, "coord must be < 2^24");
    const dim3 grid((unsigned)((n_chains + 255) / 256)), block(256);
    if (math_mode == MCMC_MATH_FAST)
        hipLaunchKernelGGL(draws_kernel<MCMC_MATH_FAST>, grid, block, 0, (hipStream_t)stream, seed, chain_offset, n_chains,
                           coord, step_offset, n_steps, normals, uniforms, angle_u0, angle_u1, sab_normals, sab_uniforms);
    else
        hipLaunchKernelGGL(draws_kernel<MCMC_MATH_PRECISE>, grid, block, 0, (hipStream_t)stream, seed, chain_offset,
                           n_chains, coord, step_offset, n_steps, normals, uniforms, angle_u0, angle_u1, sab_normals,
                           sab_uniforms);
    MCMC_HIP_CHECK(hipGetLastError());
    return MCMC_OK;
}
