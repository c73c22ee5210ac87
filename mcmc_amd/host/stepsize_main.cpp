// stepsize -- the step-size study of the reference's stepsize executable
// (stepsize/src/main.cpp:7-36) on the GPU, through the C ABI only.
//
// For every proposal width s in the sweep: n_chains independent chains of
// `iterations` MH steps on the target PDF() (r^2 Woods-Saxon(2.0, 0.54),
// stepsize/include/metropolis.hpp:19-23, or any of the Distributions.cuh
// callables), then acceptance ratio and autocorrelation at lags 1..K pooled over
// chains.  Writes statistics.txt in the reference's format
// (stepsize/include/fileio.hpp:19-43) and efficiency.txt with the integrated
// autocorrelation time 1 + 2 sum rho_k and efficiency 1/tau
// (stepsize/graphing/StepSizeAnalysis.ipynb cell 1).
//
// Defaults reproduce the reference run (widths 0.01 .. 5.00, lags 1..49, x0 = 1) with
// BASELINE config 0's shape: 1024 chains x 1e4 steps, 64-bit CPU-rule arithmetic.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mcmc_hip.h"
#include "output.hpp"

namespace {

struct Options {
    uint64_t chains = 1024, iterations = 10000, warmup = 0, seed = 314159265ull;
    uint32_t coords = 1, lags = 49;
    int first = 1, last = 500, stride = 1; // width = k / 100., main.cpp:14-17
    int precision = 64, rule = MCMC_ACCEPT_SHORTCIRCUIT, math = MCMC_MATH_PRECISE;
    mcmc_target target = {MCMC_TARGET_R2WS, 2.0f, 0.54f};
    double x0 = 1.0; // metropolis.hpp:29
    std::string outdir = "./";
    bool samples = false;
    std::vector<int> devices; // --devices 0,1,...: the chains partitioned over these GPUs (mcmc_stepsize_run_multi)
    int reduce = MCMC_REDUCE_AUTO;
};

void usage()
{
    std::puts("usage: stepsize [--chains N] [--iterations N] [--warmup N] [--coords N] [--lags K<=49]\n"
              "                [--first K] [--last K] [--stride K]   widths K/100 (default 1..500)\n"
              "                [--precision 32|64] [--rule recip|shortcircuit|nodiv|log|cross] [--math precise|fast]\n"
              "                [--target ws|r2ws|gauss|r2gauss] [--p0 X] [--p1 X] [--x0 X] [--seed S]\n"
              "                [--outdir DIR] [--samples]   (samples_<k>.txt of chain 0, needs --chains 1)\n"
              "                [--devices 0,1,...] [--reduce auto|host|rccl]   chains partitioned over GPUs, one all-reduce per width");
}

bool parse(int argc, char **argv, Options &o)
{
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](const char *name) -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "stepsize: %s needs a value\n", name);
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--help" || a == "-h") return false;
        else if (a == "--chains") o.chains = std::strtoull(val("--chains"), nullptr, 10);
        else if (a == "--iterations") o.iterations = std::strtoull(val("--iterations"), nullptr, 10);
        else if (a == "--warmup") o.warmup = std::strtoull(val("--warmup"), nullptr, 10);
        else if (a == "--coords") o.coords = (uint32_t)std::atoi(val("--coords"));
        else if (a == "--lags") o.lags = (uint32_t)std::atoi(val("--lags"));
        else if (a == "--first") o.first = std::atoi(val("--first"));
        else if (a == "--last") o.last = std::atoi(val("--last"));
        else if (a == "--stride") o.stride = std::atoi(val("--stride"));
        else if (a == "--precision") o.precision = std::atoi(val("--precision"));
        else if (a == "--seed") o.seed = std::strtoull(val("--seed"), nullptr, 10);
        else if (a == "--x0") o.x0 = std::atof(val("--x0"));
        else if (a == "--p0") o.target.p0 = (float)std::atof(val("--p0"));
        else if (a == "--p1") o.target.p1 = (float)std::atof(val("--p1"));
        else if (a == "--outdir") o.outdir = val("--outdir");
        else if (a == "--samples") o.samples = true;
        else if (a == "--devices") {
            const std::string v = val("--devices");
            size_t pos = 0;
            while (pos <= v.size()) {
                const size_t c = v.find(',', pos);
                const std::string tok = v.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
                if (!tok.empty()) o.devices.push_back(std::atoi(tok.c_str()));
                if (c == std::string::npos) break;
                pos = c + 1;
            }
        } else if (a == "--reduce") {
            const std::string r = val("--reduce");
            o.reduce = r == "host" ? MCMC_REDUCE_HOST : r == "rccl" ? MCMC_REDUCE_RCCL : MCMC_REDUCE_AUTO;
        }
        else if (a == "--math") o.math = std::strcmp(val("--math"), "fast") == 0 ? MCMC_MATH_FAST : MCMC_MATH_PRECISE;
        else if (a == "--rule") {
            const std::string r = val("--rule");
            o.rule = r == "recip" ? MCMC_ACCEPT_RECIP : r == "nodiv" ? MCMC_ACCEPT_NODIV : r == "log" ? MCMC_ACCEPT_LOG
                     : r == "cross" ? MCMC_ACCEPT_CROSS : MCMC_ACCEPT_SHORTCIRCUIT;
        } else if (a == "--target") {
            const std::string t = val("--target");
            o.target.kind = t == "ws" ? MCMC_TARGET_WS : t == "gauss" ? MCMC_TARGET_GAUSS
                            : t == "r2gauss" ? MCMC_TARGET_R2GAUSS : MCMC_TARGET_R2WS;
            if (t == "gauss" || t == "r2gauss") o.target.p0 = 1.0f, o.target.p1 = 0.0f;
        } else {
            std::fprintf(stderr, "stepsize: unknown option %s\n", a.c_str());
            return false;
        }
    }
    if (!o.outdir.empty() && o.outdir.back() != '/') o.outdir += '/';
    return true;
}

} // namespace

int main(int argc, char **argv)
{
    Options o;
    if (!parse(argc, argv, o)) {
        usage();
        return 2;
    }
    if (o.samples && (o.chains != 1 || o.coords != 1)) { // the sample buffer below holds `iterations` values
        std::fprintf(stderr, "stepsize: --samples writes the reference's one-chain files and needs --chains 1 --coords 1\n");
        return 2;
    }
    if (mcmc_device_count() <= 0) {
        std::fprintf(stderr, "stepsize: no HIP device (%s); this build has no CPU path\n", mcmc_last_error());
        return 3;
    }
    std::printf("%s: %llu chains x %llu iterations x %u coordinate(s), widths %g..%g, lags 1..%u, %d-bit\n", mcmc_version(),
                (unsigned long long)o.chains, (unsigned long long)o.iterations, o.coords, o.first / 100., o.last / 100.,
                o.lags, o.precision);

    std::vector<std::vector<double>> statistics, efficiency;
    std::vector<mcmc_chain_summary> summary(o.coords);
    std::vector<double> rho((size_t)o.coords * (o.lags ? o.lags : 1));
    std::vector<double> samples;
    if (o.samples) samples.resize(o.iterations);
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t proposals = 0;
    for (int k = o.first; k <= o.last; k += o.stride) {
        if (k % 10 == 0) std::printf("%d\n", k); // main.cpp:15
        const double width = k / 100.;           // main.cpp:17
        std::vector<float> samples32;
        void *sp = nullptr;
        if (o.samples) {
            if (o.precision == 64) sp = samples.data();
            else {
                samples32.resize(o.iterations);
                sp = samples32.data();
            }
        }
        const int rc = o.devices.empty()
                           ? mcmc_stepsize_run(&o.target, o.precision, o.rule, o.math, width, o.chains, o.coords, o.warmup, o.iterations,
                                               o.lags, o.seed, o.x0, summary.data(), rho.data(), sp)
                           : mcmc_stepsize_run_multi(&o.target, o.precision, o.rule, o.math, width, o.chains, o.coords, o.warmup,
                                                     o.iterations, o.lags, o.seed, o.x0, o.devices.data(), (int)o.devices.size(),
                                                     o.reduce, summary.data(), rho.data(), sp);
        if (rc != MCMC_OK) {
            std::fprintf(stderr, "stepsize: mcmc_stepsize_run failed (%d): %s\n", rc, mcmc_last_error());
            return 1;
        }
        proposals += o.chains * o.coords * (o.warmup + o.iterations);
        if (o.samples) {
            if (o.precision != 64) samples.assign(samples32.begin(), samples32.end());
            if (!mcmc_host::write_samples(samples, o.outdir + "samples_" + std::to_string(k) + ".txt")) { // main.cpp:21
                std::fprintf(stderr, "stepsize: cannot write samples to %s\n", o.outdir.c_str());
                return 1;
            }
        }
        // coordinate 0 is the reference's row; further coordinates are independent replicas of the same target
        std::vector<double> row = {width, summary[0].acceptance};      // main.cpp:23-25
        for (uint32_t l = 0; l < o.lags; ++l) row.push_back(rho[l]);    // main.cpp:27-30
        statistics.push_back(row);
        double tau = 0, acc = 0;
        for (uint32_t c = 0; c < o.coords; ++c) tau += summary[c].tau_int, acc += summary[c].acceptance;
        tau /= o.coords, acc /= o.coords;
        efficiency.push_back({width, acc, tau, 1.0 / tau, summary[0].mean, summary[0].variance});
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!mcmc_host::write_statistics(statistics, o.outdir + "statistics.txt")) { // main.cpp:35
        std::fprintf(stderr, "stepsize: cannot write %sstatistics.txt\n", o.outdir.c_str());
        return 1;
    }
    {
        std::ofstream f(o.outdir + "efficiency.txt");
        f << "# Stepsize Acceptance tau_int efficiency mean variance\n";
        for (const auto &r : efficiency) {
            for (double v : r) f << v << " ";
            f << "\n";
        }
    }
    size_t best = 0;
    for (size_t i = 1; i < efficiency.size(); ++i)
        if (efficiency[i][3] > efficiency[best][3]) best = i;
    std::printf("max efficiency %.4f at width %.2f (acceptance %.4f); %.3g proposals in %.2f s (%.3g /s incl. setup)\n",
                efficiency[best][3], efficiency[best][0], efficiency[best][1], (double)proposals, secs, proposals / secs);
    return 0;
}
