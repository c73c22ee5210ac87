// capi.hip -- Layer B of the C ABI: the handle API with the drivers' parameter
// names (xsection/src/main.cu:38-82), mcmc_stepsize_run, and the host
// statistics with the reference's arithmetic (Statistics.cuh, statistics.hpp,
// XSectionAnalysis.py).  Built on Layer A; owns its device memory.
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../host/output.hpp"
#include "handle.hpp"
#include "host_common.hpp"

#pragma clang fp contract(off)

using namespace mcmc;

// ---------------------------------------------------------------------
// Host statistics.
// Statistics.cuh:7-20
extern "C" float mcmc_Sum(const float *samples, int n)
{
    float res = 0.0f, err = 0.0f;
    for (int i = 0; i < n; i++) {
        const float y = samples[i] - err;
        const float t = res + y;
        err = (t - res) - y;
        res = t;
    }
    return res - err;
}

// Statistics.cuh:23-25
extern "C" float mcmc_Average(const float *samples, int n) { return mcmc_Sum(samples, n) / n; }

// Statistics.cuh:28-42 (population variance)
extern "C" float mcmc_Variance(const float *samples, float avg, int n)
{
    float res = 0.0f, err = 0.0f;
    for (int i = 0; i < n; i++) {
        float y = samples[i] - avg;
        y = y * y - err;
        const float t = res + y;
        err = (t - res) - y;
        res = t;
    }
    return res / n;
}

// Statistics.cuh:63-87
extern "C" float mcmc_AutoCorrelation(const float *samples, float avg, int lag, int n)
{
    float num = 0.0f, den = 0.0f, err = 0.0f;
    for (int i = 0; i < n - lag; i++) {
        const float y = (samples[i] - avg) * (samples[i + lag] - avg) - err;
        const float t = num + y;
        err = (t - num) - y;
        num = t;
    }
    err = 0.0f;
    for (int i = 0; i < n; i++) {
        const float y = (samples[i] - avg) * (samples[i] - avg) - err;
        const float t = den + y;
        err = (t - den) - y;
        den = t;
    }
    return num / den;
}

// Statistics.cuh:45-60: Variance(float*, int) averages first, then the form above
extern "C" float mcmc_Variance_n(const float *samples, int n) { return mcmc_Variance(samples, mcmc_Average(samples, n), n); }

// Statistics.cuh:90-115: AutoCorrelation(float*, int lag, int n) likewise
extern "C" float mcmc_AutoCorrelation_n(const float *samples, int lag, int n)
{
    return mcmc_AutoCorrelation(samples, mcmc_Average(samples, n), lag, n);
}

// statistics.hpp:6-15
extern "C" double mcmc_Average_f64(const double *nums, int n)
{
    double sum = 0;
    for (int i = 0; i < n; i++) sum += nums[i];
    return sum / n;
}

// statistics.hpp:17-34
extern "C" double mcmc_Variance_f64(const double *nums, int n)
{
    const double average = mcmc_Average_f64(nums, n);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double temp = nums[i] - average;
        sum += temp * temp;
    }
    return sum / (n - 1);
}

// statistics.hpp:36-57
extern "C" double mcmc_AutoCorrelation_f64(const double *nums, int n, int lag)
{
    const double average = mcmc_Average_f64(nums, n);
    double var = 0;
    for (int i = 0; i < n; i++) {
        const double temp = nums[i] - average;
        var += temp * temp;
    }
    var /= (n - 1);
    double sum = 0;
    for (int i = 0; i < n - lag; i++) sum += (nums[i] - average) * (nums[i + lag] - average);
    return sum / (n - lag - 1) / var;
}

// XSectionAnalysis.py:3-28
namespace {
double trapz(const double *x, const std::vector<double> &y, int n)
{
    double s = 0;
    for (int i = 0; i + 1 < n; i++) s += (y[i] * 0.5 + y[i + 1] * 0.5) * (x[i + 1] - x[i]);
    return s;
}
double trapz_unc(const double *x, const std::vector<double> &e, int n)
{
    double s = 0;
    for (int i = 0; i + 1 < n; i++) {
        const double t = std::sqrt(e[i] * e[i] * 0.5 + e[i + 1] * e[i + 1] * 0.5) * (x[i + 1] - x[i]) / 2;
        s += t * t;
    }
    return std::sqrt(s);
}
} // namespace

extern "C" int mcmc_cross_sections(const double *b, const double *S, const double *Serr, int n, double *out)
{
    if (!b || !S || !Serr || !out || n < 2) return set_error(MCMC_EINVAL, "mcmc_cross_sections needs n >= 2 points");
    const double pi = 3.14159265358979323846;
    std::vector<double> y(n), e(n);
    for (int i = 0; i < n; i++) { y[i] = b[i] * (1 - S[i]); e[i] = b[i] * Serr[i]; }
    out[0] = 10 * 2 * pi * 2 * trapz(b, y, n);
    out[1] = 10 * trapz_unc(b, e, n);
    for (int i = 0; i < n; i++) { y[i] = b[i] * (1 - S[i]) * (1 - S[i]); e[i] = b[i] * std::sqrt(2.0) * (1 - S[i]) * Serr[i]; }
    out[2] = 10 * 2 * pi * trapz(b, y, n);
    out[3] = 10 * trapz_unc(b, e, n);
    for (int i = 0; i < n; i++) { y[i] = b[i] * (1 - S[i] * S[i]); e[i] = b[i] * std::sqrt(2.0) * S[i] * Serr[i]; }
    out[4] = 10 * 2 * pi * trapz(b, y, n);
    out[5] = 10 * trapz_unc(b, e, n);
    return MCMC_OK;
}

// ---------------------------------------------------------------------
// Isotope names -> (Z, A): NuclearZAndA, xsection/tools/MCMCMacro.py:63-83.  Accepts
// "PB-208" and "208-PB", any case.  (The reference's table spells californium "CT";
// both "CF" and "CT" are accepted for Z = 98.)
namespace {
const char *const kElements[] = {
    "H",  "HE", "LI", "BE", "B",  "C",  "N",  "O",  "F",  "NE", "NA", "MG", "AL", "SI", "P",  "S",  "CL", "AR", "K",  "CA",
    "SC", "TI", "V",  "CR", "MN", "FE", "CO", "NI", "CU", "ZN", "GA", "GE", "AS", "SE", "BR", "KR", "RB", "SR", "Y",  "ZR",
    "NB", "MO", "TC", "RU", "RH", "PD", "AG", "CD", "IN", "SN", "SB", "TE", "I",  "XE", "CS", "BA", "LA", "CE", "PR", "ND",
    "PM", "SM", "EU", "GD", "TB", "DY", "HO", "ER", "TM", "YB", "LU", "HF", "TA", "W",  "RE", "OS", "IR", "PT", "AU", "HG",
    "TL", "PB", "BI", "PO", "AT", "RN", "FR", "RA", "AC", "TH", "PA", "U",  "NP", "PU", "AM", "CM", "BK", "CF", "ES", "FM",
    "MD", "NO", "LR", "RF", "DB", "SG", "BH", "HS", "MT", "DS", "RG", "CN", "NH", "FL", "MC", "LV", "TS", "OG"};
}

extern "C" int mcmc_isotope(const char *name, int *Z, int *A)
{
    if (!name || !Z || !A) return set_error(MCMC_EINVAL, "null pointer");
    const char *dash = std::strchr(name, '-');
    if (!dash || dash == name || dash[1] == 0) return set_error(MCMC_EINVAL, "isotope must look like PB-208 or 208-PB");
    std::string left(name, dash), right(dash + 1);
    const bool digits_first = left[0] >= '0' && left[0] <= '9';
    std::string sym = digits_first ? right : left;
    const std::string num = digits_first ? left : right;
    for (char &c : sym) c = (char)std::toupper((unsigned char)c);
    if (sym == "CT") sym = "CF";
    char *end = nullptr;
    const long a = std::strtol(num.c_str(), &end, 10);
    if (*end != 0 || a <= 0) return set_error(MCMC_EINVAL, "bad mass number in isotope name");
    for (size_t i = 0; i < sizeof(kElements) / sizeof(kElements[0]); ++i) {
        if (sym == kElements[i]) {
            if (a < (long)i + 1) return set_error(MCMC_EINVAL, "mass number smaller than the atomic number");
            *Z = (int)i + 1;
            *A = (int)a;
            return MCMC_OK;
        }
    }
    return set_error(MCMC_EINVAL, "unknown element symbol");
}

// ---------------------------------------------------------------------
// Handle (struct mcmc_handle_s: handle.hpp).
extern "C" void mcmc_config_default(mcmc_config *c)
{
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->nNeutronA = c->nProtonA = c->nNeutronB = c->nProtonB = 2;   // main.cu:38-41
    const mcmc_target pdf = {MCMC_TARGET_R2WS, 2.f, 0.54f};         // main.cu:14-19
    c->neutronA = c->protonA = c->neutronB = c->protonB = pdf;
    c->totalBlocks = 1 << 8;        // main.cu:54
    c->threadsPerBlock = 1 << 7;    // main.cu:55
    c->samplesPerThread = 1 << 8;   // main.cu:56
    c->totalRuns = 1 << 4;          // main.cu:58
    c->samplesForMix = 1 << 13;     // main.cu:60
    c->b0 = 0;                      // main.cu:63
    c->db = 0.1f;                   // main.cu:64
    c->ndbs = 200;                  // main.cu:66
    c->seed = 314159265ull;         // MCMCMacro.py:103 (main.cu:69 seeds from the clock)
    c->stepsize = 2.2f;             // main.cu:149
    c->initSample = 1.0f;           // main.cu:132
    c->sigmaNN = 4.3f;              // IntegrateSAB.cuh:11
    c->beta = 0.2f;                 // IntegrateSAB.cuh:10
    c->accept_rule = MCMC_ACCEPT_RECIP;
    c->math_mode = MCMC_MATH_FAST;
    c->device = 0;
    c->event_offset = 0;
    c->integrateRMS = 1;            // main.cu:81
    c->integrateSAB = 1;            // main.cu:82
    c->outDir = nullptr;            // main.cu:73 is "../data/"; the library default is the working directory
    c->outFile = nullptr;           // main.cu:74 "out.root" -> "out" (+ ".RMS.csv" / ".S_AB.csv")
}

namespace {

void free_handle(mcmc_handle h)
{
    if (!h) return;
    if (h->d_prev) (void)hipFree(h->d_prev);
    if (h->d_targets) (void)hipFree(h->d_targets);
    if (h->d_result) (void)hipFree(h->d_result);
    if (h->d_means) (void)hipFree(h->d_means);
    if (h->d_ws) (void)hipFree(h->d_ws);
    if (h->d_scan) (void)hipFree(h->d_scan);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int ensure_scan_buffer(mcmc_handle h, size_t len)
{
    if (h->scan_len >= len) return MCMC_OK;
    MCMC_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (h->d_scan) (void)hipFree(h->d_scan);
    h->d_scan = nullptr;
    h->scan_len = 0;
    MCMC_HIP_CHECK(hipMalloc((void **)&h->d_scan, sizeof(double) * len));
    h->scan_len = len;
    return MCMC_OK;
}

} // namespace

extern "C" int mcmc_create(const mcmc_config *cfg, mcmc_handle *out)
{
    if (!cfg || !out) return set_error(MCMC_EINVAL, "null pointer");
    if (cfg->totalBlocks <= 0 || cfg->threadsPerBlock <= 0) return set_error(MCMC_EINVAL, "totalBlocks and threadsPerBlock must be > 0");
    return create_handle(cfg, (uint64_t)cfg->totalBlocks * (uint64_t)cfg->threadsPerBlock, out); // main.cu:89
}

// `events` = the handle's share of totalThreads (all of them for mcmc_create; a partition of them per GPU for mcmc_multi_create)
int mcmc::create_handle(const mcmc_config *cfg, uint64_t events, mcmc_handle *out)
{
    if (!cfg || !out) return set_error(MCMC_EINVAL, "null pointer");
    if (events == 0) return set_error(MCMC_EINVAL, "a handle needs at least one event");
    if (cfg->nNeutronA < 0 || cfg->nProtonA < 0 || cfg->nNeutronB < 0 || cfg->nProtonB < 0)
        return set_error(MCMC_EINVAL, "negative nucleon count");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_error(MCMC_ENODEV, "no HIP device");
    DeviceGuard restore_device;
    MCMC_HIP_CHECK(hipSetDevice(cfg->device));
    mcmc_handle h = new mcmc_handle_s();
    std::memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    h->nA = cfg->nNeutronA + cfg->nProtonA;
    h->nB = cfg->nNeutronB + cfg->nProtonB;
    h->nN = h->nA + h->nB;
    h->T = events;
    if (h->nN == 0) { delete h; return set_error(MCMC_EINVAL, "no nucleons"); }
    std::snprintf(h->outDir, sizeof h->outDir, "%s", cfg->outDir ? cfg->outDir : "./");
    std::snprintf(h->outFile, sizeof h->outFile, "%s", cfg->outFile ? cfg->outFile : "out");
    h->cfg.outDir = h->outDir;
    h->cfg.outFile = h->outFile;

    // per-nucleon densities in the order of TemplateBuilder::ConcatArray(pdfArrA, pdfArrB)
    // (main.cu:49-51; MCMCMacro.py:315-345 concatenates protons A, neutrons A, protons B, neutrons B)
    std::vector<mcmc_target> tg;
    if (cfg->nucleon_targets) { // one density per nucleon, already in that order (MCMCMacro.py:172-199)
        tg.assign(cfg->nucleon_targets, cfg->nucleon_targets + h->nN);
    } else {
        for (int i = 0; i < cfg->nProtonA; i++) tg.push_back(cfg->protonA);
        for (int i = 0; i < cfg->nNeutronA; i++) tg.push_back(cfg->neutronA);
        for (int i = 0; i < cfg->nProtonB; i++) tg.push_back(cfg->protonB);
        for (int i = 0; i < cfg->nNeutronB; i++) tg.push_back(cfg->neutronB);
    }
    h->cfg.nucleon_targets = nullptr; // the caller's array is not kept: the densities live in d_targets
    h->uniform_kind = tg[0].kind;
    for (const auto &t : tg) {
        if (t.kind < 0 || t.kind > 3) { delete h; return set_error(MCMC_EINVAL, "bad target kind"); }
        if (t.kind != h->uniform_kind) h->uniform_kind = -1;
    }

#define MCMC_CREATE_CHECK(expr)                                       \
    do {                                                              \
        hipError_t _e = (expr);                                       \
        if (_e != hipSuccess) {                                       \
            free_handle(h);                                           \
            return set_hip_error(_e, __FILE__, __LINE__);             \
        }                                                             \
    } while (0)
    MCMC_CREATE_CHECK(hipStreamCreate(&h->stream));
    MCMC_CREATE_CHECK(hipMalloc((void **)&h->d_prev, sizeof(float) * h->nN * h->T));        // main.cu:126
    MCMC_CREATE_CHECK(hipMalloc((void **)&h->d_targets, sizeof(mcmc_target) * h->nN));
    MCMC_CREATE_CHECK(hipMalloc((void **)&h->d_result, sizeof(float) * h->nN * h->T));      // main.cu:164
    MCMC_CREATE_CHECK(hipMalloc((void **)&h->d_means, sizeof(double) * (h->nN + 4)));
    mcmc_xs_args a = xs_args(h, cfg->stepsize);
    h->ws_bytes = mcmc_xs_workspace_bytes(&a);
    MCMC_CREATE_CHECK(hipMalloc(&h->d_ws, h->ws_bytes));
    MCMC_CREATE_CHECK(hipMemcpy(h->d_targets, tg.data(), sizeof(mcmc_target) * h->nN, hipMemcpyHostToDevice));
#undef MCMC_CREATE_CHECK
    // RandStateInit (main.cu:130) has no counterpart: the Philox counter is (event, nucleon, sweep).
    int rc = mcmc_launch_init_sample_array(h->d_prev, cfg->initSample, h->nN, h->T, h->stream); // main.cu:132
    if (rc) { free_handle(h); return rc; }
    hipError_t e = hipStreamSynchronize(h->stream); // main.cu:134
    if (e != hipSuccess) { free_handle(h); return set_hip_error(e, __FILE__, __LINE__); }
    *out = h;
    return MCMC_OK;
}

extern "C" int mcmc_destroy(mcmc_handle h)
{
    if (!h) return set_error(MCMC_EINVAL, "null handle");
    DeviceGuard restore_device;
    (void)hipSetDevice(h->cfg.device);
    free_handle(h);
    return MCMC_OK;
}

extern "C" int mcmc_wait(mcmc_handle h)
{
    if (!h) return set_error(MCMC_EINVAL, "null handle");
    MCMC_HIP_CHECK(hipStreamSynchronize(h->stream));
    return MCMC_OK;
}

extern "C" int mcmc_warmup(mcmc_handle h, float stepsize, int iterations)
{
    if (!h) return set_error(MCMC_EINVAL, "null handle");
    if (iterations < 0) return set_error(MCMC_EINVAL, "iterations must be >= 0");
    DeviceGuard restore_device;
    MCMC_HIP_CHECK(hipSetDevice(h->cfg.device));
    mcmc_xs_args a = xs_args(h, stepsize);
    int rc = mcmc_launch_warmup_metropolis(&a, (uint32_t)iterations, h->stream);
    if (rc) return rc;
    h->sweeps_done += (uint64_t)iterations;
    MCMC_HIP_CHECK(hipStreamSynchronize(h->stream)); // main.cu:151
    return MCMC_OK;
}

extern "C" int mcmc_init_from_pdf(mcmc_handle h)
{
    if (!h) return set_error(MCMC_EINVAL, "null handle");
    DeviceGuard restore_device;
    MCMC_HIP_CHECK(hipSetDevice(h->cfg.device));
    mcmc_xs_args a = xs_args(h, h->cfg.stepsize);
    return mcmc_launch_init_sample_from_pdf(&a, h->stream);
}

extern "C" int mcmc_integrate_r2(mcmc_handle h, float stepsize, int samples, float *out_run_mean)
{
    if (!h || !out_run_mean) return set_error(MCMC_EINVAL, "null pointer");
    if (samples <= 0) return set_error(MCMC_EINVAL, "samples must be > 0");
    DeviceGuard restore_device;
    MCMC_HIP_CHECK(hipSetDevice(h->cfg.device));
    mcmc_xs_args a = xs_args(h, stepsize);
    int rc = mcmc_launch_integrate_r2(&a, h->d_result, (uint32_t)samples, h->d_means, h->stream);
    if (rc) return rc;
    h->sweeps_done += (uint64_t)samples;
    std::vector<double> m(h->nN);
    MCMC_HIP_CHECK(hipMemcpyAsync(m.data(), h->d_means, sizeof(double) * h->nN, hipMemcpyDeviceToHost, h->stream));
    MCMC_HIP_CHECK(hipStreamSynchronize(h->stream));
    for (int n = 0; n < h->nN; n++) out_run_mean[n] = (float)m[n];
    return MCMC_OK;
}

extern "C" int mcmc_integrate_sab(mcmc_handle h, float stepsize, int samples, float b, float *out_run_mean, double *hits4,
                                  float dmax2)
{
    if (!h || !out_run_mean) return set_error(MCMC_EINVAL, "null pointer");
    if (samples <= 0) return set_error(MCMC_EINVAL, "samples must be > 0");
    DeviceGuard restore_device;
    MCMC_HIP_CHECK(hipSetDevice(h->cfg.device));
    mcmc_xs_args a = xs_args(h, stepsize);
    int rc = mcmc_launch_integrate_sab(&a, h->d_result, (uint32_t)samples, b, h->d_means, hits4 ? h->d_means + 1 : nullptr,
                                       dmax2, h->stream);
    if (rc) return rc;
    h->sweeps_done += (uint64_t)samples;
    double m[5];
    MCMC_HIP_CHECK(hipMemcpyAsync(m, h->d_means, sizeof(double) * (hits4 ? 5 : 1), hipMemcpyDeviceToHost, h->stream));
    MCMC_HIP_CHECK(hipStreamSynchronize(h->stream));
    *out_run_mean = (float)m[0];
    if (hits4) std::memcpy(hits4, m + 1, sizeof(double) * 4);
    return MCMC_OK;
}

// main.cu:157-286.  The reference double-buffers launches against D2H copies and host averages
// (main.cu:197-240); here every run's per-nucleon mean is reduced on the device into
// d_scan[run][nN], all runs are queued without a host round trip, and ONE copy ends the loop.
extern "C" int mcmc_run_rms(mcmc_handle h, float *rms, float *rms_stderr)
{
    if (!h || !rms || !rms_stderr) return set_error(MCMC_EINVAL, "null pointer");
    const int R = h->cfg.totalRuns;
    if (R <= 0) return set_error(MCMC_EINVAL, "totalRuns must be > 0");
    if (h->cfg.samplesPerThread <= 0) return set_error(MCMC_EINVAL, "samplesPerThread must be > 0");
    DeviceGuard restore_device;
    MCMC_HIP_CHECK(hipSetDevice(h->cfg.device));
    int rc = ensure_scan_buffer(h, (size_t)R * h->nN);
    if (rc) return rc;
    for (int run = 0; run < R; run++) {
        mcmc_xs_args a = xs_args(h, h->cfg.stepsize);
        rc = mcmc_launch_integrate_r2(&a, h->d_result, (uint32_t)h->cfg.samplesPerThread, h->d_scan + (size_t)run * h->nN,
                                      h->stream);
        if (rc) return rc;
        h->sweeps_done += (uint64_t)h->cfg.samplesPerThread;
    }
    std::vector<double> m((size_t)R * h->nN);
    MCMC_HIP_CHECK(hipMemcpyAsync(m.data(), h->d_scan, sizeof(double) * m.size(), hipMemcpyDeviceToHost, h->stream));
    MCMC_HIP_CHECK(hipStreamSynchronize(h->stream));
    std::vector<float> ra(R);
    for (int n = 0; n < h->nN; n++) { // main.cu:221-223,253-258
        for (int run = 0; run < R; run++) ra[run] = (float)m[(size_t)run * h->nN + n];
        const float finalR2Average = mcmc_Average(ra.data(), R);
        const float finalR2Stderr = std::sqrt(mcmc_Variance(ra.data(), finalR2Average, R) / R);
        rms[n] = std::sqrt(finalR2Average);
        rms_stderr[n] = finalR2Stderr / (2 * rms[n]);
    }
    return MCMC_OK;
}

// main.cu:289-413.  All ndbs x totalRuns launches are queued on the handle's stream, each writing its
// run mean (and hit counters) into d_scan[dbs][run][1 + 4]; one copy and one synchronisation at the end.
extern "C" int mcmc_run_sab_scan(mcmc_handle h, float *b_out, float *S_AB, float *S_AB_stderr, double *p_inel, float dmax2)
{
    if (!h || !b_out || !S_AB || !S_AB_stderr) return set_error(MCMC_EINVAL, "null pointer");
    const int R = h->cfg.totalRuns, nb = h->cfg.ndbs;
    if (R <= 0) return set_error(MCMC_EINVAL, "totalRuns must be > 0");
    if (nb <= 0) return set_error(MCMC_EINVAL, "ndbs must be > 0");
    if (h->cfg.samplesPerThread <= 0) return set_error(MCMC_EINVAL, "samplesPerThread must be > 0");
    DeviceGuard restore_device;
    MCMC_HIP_CHECK(hipSetDevice(h->cfg.device));
    int rc = ensure_scan_buffer(h, (size_t)nb * R * 5);
    if (rc) return rc;
    for (int dbs = 0; dbs < nb; dbs++) {
        const float b = h->cfg.b0 + dbs * h->cfg.db; // main.cu:328
        for (int run = 0; run < R; run++) {
            double *slot = h->d_scan + ((size_t)dbs * R + run) * 5;
            mcmc_xs_args a = xs_args(h, h->cfg.stepsize);
            rc = mcmc_launch_integrate_sab(&a, h->d_result, (uint32_t)h->cfg.samplesPerThread, b, slot,
                                           p_inel ? slot + 1 : nullptr, dmax2, h->stream);
            if (rc) return rc;
            h->sweeps_done += (uint64_t)h->cfg.samplesPerThread;
        }
    }
    std::vector<double> m((size_t)nb * R * 5);
    MCMC_HIP_CHECK(hipMemcpyAsync(m.data(), h->d_scan, sizeof(double) * m.size(), hipMemcpyDeviceToHost, h->stream));
    MCMC_HIP_CHECK(hipStreamSynchronize(h->stream));
    std::vector<float> runAverages(R);
    for (int dbs = 0; dbs < nb; dbs++) {
        double inel = 0, nsamp = 0;
        for (int run = 0; run < R; run++) {
            const double *slot = &m[((size_t)dbs * R + run) * 5];
            runAverages[run] = (float)slot[0]; // main.cu:359
            if (p_inel) { inel += slot[2]; nsamp += slot[4]; }
        }
        const float finalAverage = mcmc_Average(runAverages.data(), R);                  // main.cu:384
        const float finalVariance = mcmc_Variance(runAverages.data(), finalAverage, R);  // main.cu:385
        b_out[dbs] = h->cfg.b0 + dbs * h->cfg.db;
        S_AB[dbs] = finalAverage;
        S_AB_stderr[dbs] = std::sqrt(finalVariance / R);                                 // main.cu:386
        if (p_inel) p_inel[dbs] = nsamp > 0 ? inel / nsamp : 0.0;
    }
    return MCMC_OK;
}

// The b scan of main.cu:289-413 for the black-disk inelastic probability alone (MODE 4 kernel).
extern "C" int mcmc_run_inel_scan(mcmc_handle h, float *b_out, double *p_inel, double *p_inel_stderr, float dmax2)
{
    if (!h || !b_out || !p_inel) return set_error(MCMC_EINVAL, "null pointer");
    const int R = h->cfg.totalRuns, nb = h->cfg.ndbs;
    if (R <= 0) return set_error(MCMC_EINVAL, "totalRuns must be > 0");
    if (nb <= 0) return set_error(MCMC_EINVAL, "ndbs must be > 0");
    if (h->cfg.samplesPerThread <= 0) return set_error(MCMC_EINVAL, "samplesPerThread must be > 0");
    DeviceGuard restore_device;
    MCMC_HIP_CHECK(hipSetDevice(h->cfg.device));
    int rc = ensure_scan_buffer(h, (size_t)nb * R * 2);
    if (rc) return rc;
    for (int dbs = 0; dbs < nb; dbs++) {
        const float b = h->cfg.b0 + dbs * h->cfg.db; // main.cu:328
        for (int run = 0; run < R; run++) {
            mcmc_xs_args a = xs_args(h, h->cfg.stepsize);
            rc = mcmc_launch_integrate_inel(&a, (uint32_t)h->cfg.samplesPerThread, b, h->d_scan + ((size_t)dbs * R + run) * 2,
                                            dmax2, h->stream);
            if (rc) return rc;
            h->sweeps_done += (uint64_t)h->cfg.samplesPerThread;
        }
    }
    std::vector<double> m((size_t)nb * R * 2);
    MCMC_HIP_CHECK(hipMemcpyAsync(m.data(), h->d_scan, sizeof(double) * m.size(), hipMemcpyDeviceToHost, h->stream));
    MCMC_HIP_CHECK(hipStreamSynchronize(h->stream));
    std::vector<double> pr(R);
    for (int dbs = 0; dbs < nb; dbs++) {
        for (int run = 0; run < R; run++) {
            const double *slot = &m[((size_t)dbs * R + run) * 2];
            pr[run] = slot[1] > 0 ? slot[0] / slot[1] : 0.0;
        }
        const double avg = mcmc_Average_f64(pr.data(), R);
        b_out[dbs] = h->cfg.b0 + dbs * h->cfg.db;
        p_inel[dbs] = avg;
        if (p_inel_stderr) p_inel_stderr[dbs] = R > 1 ? std::sqrt(mcmc_Variance_f64(pr.data(), R) / R) : 0.0;
    }
    return MCMC_OK;
}

// main.cu:157-416 as the config describes it: RMS integration (:157-286) when integrateRMS, the S_AB scan (:289-413)
// when integrateSAB, results written with the TTree column names of IntegrationOutputHandler.cuh:83-112 as CSV.
extern "C" int mcmc_run_all(mcmc_handle h, double *sigma6)
{
    if (!h) return set_error(MCMC_EINVAL, "null handle");
    std::string dir = h->outDir;
    if (!dir.empty() && dir.back() != '/') dir += '/';
    const std::string base = dir + h->outFile;
    if (h->cfg.integrateRMS) {
        std::vector<float> rms(h->nN), err(h->nN);
        const int rc = mcmc_run_rms(h, rms.data(), err.data());
        if (rc) return rc;
        if (!mcmc_host::write_rms_csv(rms, err, h->nA, base + ".RMS.csv")) return set_error(MCMC_EINVAL, "cannot write <outDir><outFile>.RMS.csv");
    }
    if (h->cfg.integrateSAB) {
        const int n = h->cfg.ndbs;
        if (n <= 0) return set_error(MCMC_EINVAL, "ndbs must be > 0");
        std::vector<float> b(n), S(n), E(n);
        const int rc = mcmc_run_sab_scan(h, b.data(), S.data(), E.data(), nullptr, 0.0f);
        if (rc) return rc;
        if (!mcmc_host::write_sab_csv(b, S, E, nullptr, base + ".S_AB.csv")) return set_error(MCMC_EINVAL, "cannot write <outDir><outFile>.S_AB.csv");
        if (sigma6 && n >= 2) {
            std::vector<double> bd(b.begin(), b.end()), Sd(S.begin(), S.end()), Ed(E.begin(), E.end());
            const int rc2 = mcmc_cross_sections(bd.data(), Sd.data(), Ed.data(), n, sigma6);
            if (rc2) return rc2;
        }
    }
    return MCMC_OK;
}

extern "C" int mcmc_get_state(mcmc_handle h, float *host_out)
{
    if (!h || !host_out) return set_error(MCMC_EINVAL, "null pointer");
    DeviceGuard restore_device;
    MCMC_HIP_CHECK(hipSetDevice(h->cfg.device));
    MCMC_HIP_CHECK(hipStreamSynchronize(h->stream));
    MCMC_HIP_CHECK(hipMemcpy(host_out, h->d_prev, sizeof(float) * h->nN * h->T, hipMemcpyDeviceToHost));
    return MCMC_OK;
}
