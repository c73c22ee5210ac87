// xsection -- the Glauber integration driver of the reference
// (xsection/src/main.cu:31-424) over the C ABI of libmcmc_hip.so.
//
// Same flow: allocate + InitSampleArray (main.cu:123-134) -> WarmupMetropolis
// (:149) -> RMS integration (:157-286) -> S_AB scan over impact parameters
// (:289-413) -> write results (:416).  The reference fixes every parameter at
// compile time (main.cu:38-82 or a generated .mac file, MCMCMacro.py:305-368);
// here the SAME names are run-time `name=value` arguments:
//
//   nNeutronA nProtonA nNeutronB nProtonB | isotopeA=PB-208 isotopeB=AU-197
//   pdfA=r2ws:6.62:0.546  pdfB=...  (kind:p0[:p1], kinds ws r2ws gauss r2gauss; both nuclei: pdf=...)
//   totalBlocks threadsPerBlock samplesPerThread totalRuns samplesForMix
//   b0 db ndbs seed stepsize sigmaNN beta outDir outFile integrateRMS integrateSAB
//   blackDisk=1 (adds the hit-test inelastic probability, d < sqrt(sigmaNN/pi))  math=fast|precise
//   macro=<file>  a macro file written by the reference's generator (MCMCMacro.py:285-372; main.cu:22-37 compiles one
//   executable per macro): its constants and per-nucleon densities are read at run time (macro.hpp); arguments after
//   it override it.  printConfig=1 prints the effective configuration and exits without touching the GPU.
//   devices=0,1,...  reduce=auto|host|rccl : the events (totalBlocks x threadsPerBlock) partitioned over these GPUs, one
//   host thread and stream per GPU, ONE all-reduce of the scan's sufficient statistics (mcmc_multi_*, include/mcmc_hip.h
//   Layer C); results are bit-identical for any device list.  Without devices=: the one-GPU handle on `device`.
//
// Results: <outDir><outFile>.RMS.csv and .S_AB.csv with the reference's TTree columns
// (IntegrationOutputHandler.cuh:83-87,109-112), and sigma_tot / sigma_el / sigma_rxn from the
// S_AB table (xsection/tools/XSectionAnalysis.py:3-28) on stdout.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "macro.hpp"
#include "mcmc_hip.h"
#include "output.hpp"

namespace {

bool parse_pdf(const std::string &v, mcmc_target &t)
{
    const size_t c1 = v.find(':');
    if (c1 == std::string::npos) return false;
    const std::string kind = v.substr(0, c1);
    t.kind = kind == "ws" ? MCMC_TARGET_WS : kind == "r2ws" ? MCMC_TARGET_R2WS : kind == "gauss" ? MCMC_TARGET_GAUSS
             : kind == "r2gauss" ? MCMC_TARGET_R2GAUSS : -1;
    if (t.kind < 0) return false;
    const size_t c2 = v.find(':', c1 + 1);
    t.p0 = (float)std::atof(v.substr(c1 + 1, c2 == std::string::npos ? std::string::npos : c2 - c1 - 1).c_str());
    t.p1 = c2 == std::string::npos ? 0.0f : (float)std::atof(v.substr(c2 + 1).c_str());
    return true;
}

std::vector<int> parse_devices(const std::string &v)
{
    std::vector<int> d;
    size_t pos = 0;
    while (pos <= v.size()) {
        const size_t c = v.find(',', pos);
        const std::string tok = v.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
        if (!tok.empty()) d.push_back(std::atoi(tok.c_str()));
        if (c == std::string::npos) break;
        pos = c + 1;
    }
    return d;
}

void die(const char *what)
{
    std::fprintf(stderr, "xsection: %s: %s\n", what, mcmc_last_error());
    std::exit(1);
}

} // namespace

int main(int argc, char **argv)
{
    mcmc_config cfg;
    mcmc_config_default(&cfg); // the constants of main.cu:38-82
    std::string outDir = "./", outFile = "out"; // main.cu:73-74 ("../data/", "out.root")
    bool integrateRMS = true, integrateSAB = true, blackDisk = false, inelOnly = false; // main.cu:81-82
    bool autoStep = false, randomStart = false;
    std::vector<int> devices; // devices=...: the multi-GPU path
    int reduce = MCMC_REDUCE_AUTO;
    std::vector<mcmc_target> perNucleon; // macro=...: one density per nucleon (pdfArr order), else empty
    bool printConfig = false;

    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        const size_t eq = a.find('=');
        if (a == "--help" || a == "-h" || eq == std::string::npos) {
            if (eq == std::string::npos && a != "--help" && a != "-h" && argc == 2) { // main.cu:77-79: lone argument = output file
                outFile = a;
                continue;
            }
            std::puts("usage: xsection [name=value ...]   names: nNeutronA nProtonA nNeutronB nProtonB isotopeA isotopeB pdf pdfA\n"
                      "       pdfB totalBlocks threadsPerBlock samplesPerThread totalRuns samplesForMix b0 db ndbs seed stepsize\n"
                      "       sigmaNN beta outDir outFile integrateRMS integrateSAB blackDisk inelOnly randomStart rule math device\n"
                      "       devices=0,1,... reduce=auto|host|rccl macro=<file written by MCMCMacro.py> printConfig=1");
            return a == "--help" || a == "-h" ? 0 : 2;
        }
        const std::string k = a.substr(0, eq), v = a.substr(eq + 1);
        const int iv = std::atoi(v.c_str());
        const float fv = (float)std::atof(v.c_str());
        mcmc_target t;
        int Z, A;
        // a nucleus or density given after macro= replaces the macro's per-nucleon list by the per-species densities
        if (k == "nNeutronA" || k == "nProtonA" || k == "nNeutronB" || k == "nProtonB" || k == "isotopeA" || k == "isotopeB" ||
            k == "pdf" || k == "pdfA" || k == "pdfB")
            perNucleon.clear();
        if (k == "macro") { // IntegrationSettings.GenerateCode output, read instead of compiled (main.cu:22-37)
            mcmc_host::MacroSettings ms;
            std::string err;
            if (!mcmc_host::read_macro(v, ms, err)) {
                std::fprintf(stderr, "xsection: macro %s: %s\n", v.c_str(), err.c_str());
                return 2;
            }
            auto num = [&](const char *name, auto &dst) {
                const auto it = ms.numbers.find(name);
                if (it != ms.numbers.end()) dst = static_cast<std::decay_t<decltype(dst)>>(it->second);
            };
            num("nNeutronA", cfg.nNeutronA), num("nProtonA", cfg.nProtonA), num("nNeutronB", cfg.nNeutronB), num("nProtonB", cfg.nProtonB);
            num("totalBlocks", cfg.totalBlocks), num("threadsPerBlock", cfg.threadsPerBlock), num("samplesPerThread", cfg.samplesPerThread);
            num("totalRuns", cfg.totalRuns), num("samplesForMix", cfg.samplesForMix), num("b0", cfg.b0), num("db", cfg.db), num("ndbs", cfg.ndbs);
            num("integrateRMS", integrateRMS), num("integrateSAB", integrateSAB);
            if (ms.has_seed) cfg.seed = ms.seed;
            else if (ms.seed_from_clock) std::printf("macro: seed taken from the clock there; using %llu (seed=... overrides)\n", (unsigned long long)cfg.seed);
            if (ms.strings.count("outDir")) outDir = ms.strings["outDir"];
            if (ms.strings.count("outFile")) { // "out.root" names the reference's ROOT file; the CSV tables take its stem
                outFile = ms.strings["outFile"];
                if (outFile.size() > 5 && outFile.compare(outFile.size() - 5, 5, ".root") == 0) outFile.erase(outFile.size() - 5);
            }
            for (const char *name : {"integrateIAB", "integrateJAB"})
                if (ms.numbers.count(name) && ms.numbers[name] != 0.0)
                    std::printf("macro: %s is set; the reference's main.cu has no such integral, neither has this driver\n", name);
            for (const auto &n : ms.notes) std::printf("macro: %s\n", n.c_str());
            if (!ms.nucleons.empty()) {
                const size_t nA = (size_t)cfg.nProtonA + cfg.nNeutronA, nB = (size_t)cfg.nProtonB + cfg.nNeutronB;
                if (ms.nucleons.size() != nA + nB) {
                    std::fprintf(stderr, "xsection: macro %s: pdfArr has %zu densities for %zu + %zu nucleons\n", v.c_str(),
                                 ms.nucleons.size(), nA, nB);
                    return 2;
                }
                perNucleon = ms.nucleons;
                // the per-species fields follow the first nucleon of each group (autoStep, printConfig)
                if (cfg.nProtonA > 0) cfg.protonA = perNucleon[0];
                if (cfg.nNeutronA > 0) cfg.neutronA = perNucleon[cfg.nProtonA];
                if (cfg.nProtonB > 0) cfg.protonB = perNucleon[nA];
                if (cfg.nNeutronB > 0) cfg.neutronB = perNucleon[nA + cfg.nProtonB];
            }
        } else if (k == "printConfig") printConfig = iv != 0;
        else if (k == "nNeutronA") cfg.nNeutronA = iv;
        else if (k == "nProtonA") cfg.nProtonA = iv;
        else if (k == "nNeutronB") cfg.nNeutronB = iv;
        else if (k == "nProtonB") cfg.nProtonB = iv;
        else if (k == "isotopeA" || k == "isotopeB") { // IntegrationSettings.SetIsotopeA/B, MCMCMacro.py:145-157
            if (mcmc_isotope(v.c_str(), &Z, &A)) die(a.c_str());
            if (k == "isotopeA") cfg.nProtonA = Z, cfg.nNeutronA = A - Z;
            else cfg.nProtonB = Z, cfg.nNeutronB = A - Z;
        } else if (k == "pdf" || k == "pdfA" || k == "pdfB") {
            if (!parse_pdf(v, t)) {
                std::fprintf(stderr, "xsection: bad density '%s' (kind:p0[:p1])\n", v.c_str());
                return 2;
            }
            if (k != "pdfB") cfg.neutronA = cfg.protonA = t;
            if (k != "pdfA") cfg.neutronB = cfg.protonB = t;
        } else if (k == "totalBlocks") cfg.totalBlocks = iv;
        else if (k == "threadsPerBlock") cfg.threadsPerBlock = iv;
        else if (k == "samplesPerThread") cfg.samplesPerThread = iv;
        else if (k == "totalRuns") cfg.totalRuns = iv;
        else if (k == "samplesForMix") cfg.samplesForMix = iv;
        else if (k == "b0") cfg.b0 = fv;
        else if (k == "db") cfg.db = fv;
        else if (k == "ndbs") cfg.ndbs = iv;
        else if (k == "seed") cfg.seed = std::strtoull(v.c_str(), nullptr, 10);
        else if (k == "stepsize") { // a number, or "auto": tuned on the device (mcmc_tune_stepsize)
            autoStep = v == "auto";
            if (!autoStep) cfg.stepsize = fv;
        }
        else if (k == "sigmaNN") cfg.sigmaNN = fv;
        else if (k == "beta") cfg.beta = fv;
        else if (k == "device") cfg.device = iv;
        else if (k == "devices") devices = parse_devices(v);
        else if (k == "reduce") reduce = v == "host" ? MCMC_REDUCE_HOST : v == "rccl" ? MCMC_REDUCE_RCCL : MCMC_REDUCE_AUTO;
        else if (k == "rule") // acceptance test: the reference's p'(1/p) >= U, the reciprocal-free forms of p' >= U p, or log space
            cfg.accept_rule = v == "cross" ? MCMC_ACCEPT_CROSS : v == "nodiv" ? MCMC_ACCEPT_NODIV
                              : v == "log" ? MCMC_ACCEPT_LOG : MCMC_ACCEPT_RECIP;
        else if (k == "math") cfg.math_mode = v == "precise" ? MCMC_MATH_PRECISE : MCMC_MATH_FAST;
        else if (k == "outDir") outDir = v;
        else if (k == "outFile") outFile = v;
        else if (k == "integrateRMS") integrateRMS = iv != 0;
        else if (k == "integrateSAB") integrateSAB = iv != 0;
        else if (k == "blackDisk") blackDisk = iv != 0;
        else if (k == "inelOnly") inelOnly = iv != 0;
        else if (k == "randomStart") randomStart = iv != 0;
        else {
            std::fprintf(stderr, "xsection: unknown parameter '%s'\n", k.c_str());
            return 2;
        }
    }
    if (!outDir.empty() && outDir.back() != '/') outDir += '/';
    cfg.outDir = outDir.c_str();   // main.cu:73-82 as config fields (mcmc_run_all reads them; this driver adds the
    cfg.outFile = outFile.c_str(); // black-disk / inelastic-only / tuned-width options around the same calls)
    cfg.integrateRMS = integrateRMS;
    cfg.integrateSAB = integrateSAB;
    cfg.nucleon_targets = perNucleon.empty() ? nullptr : perNucleon.data();
    if (printConfig) { // the effective configuration, one name per line; densities as runs of equal neighbours in pdfArr order
        std::printf("nNeutronA %d\nnProtonA %d\nnNeutronB %d\nnProtonB %d\n", cfg.nNeutronA, cfg.nProtonA, cfg.nNeutronB, cfg.nProtonB);
        std::printf("totalBlocks %d\nthreadsPerBlock %d\nsamplesPerThread %d\ntotalRuns %d\nsamplesForMix %d\n", cfg.totalBlocks,
                    cfg.threadsPerBlock, cfg.samplesPerThread, cfg.totalRuns, cfg.samplesForMix);
        std::printf("b0 %.9g\ndb %.9g\nndbs %d\nseed %llu\nstepsize %.9g\nsigmaNN %.9g\nbeta %.9g\n", cfg.b0, cfg.db, cfg.ndbs,
                    (unsigned long long)cfg.seed, cfg.stepsize, cfg.sigmaNN, cfg.beta);
        std::printf("outDir %s\noutFile %s\nintegrateRMS %d\nintegrateSAB %d\n", outDir.c_str(), outFile.c_str(), (int)integrateRMS,
                    (int)integrateSAB);
        std::vector<mcmc_target> all(perNucleon);
        if (all.empty()) {
            all.insert(all.end(), (size_t)cfg.nProtonA, cfg.protonA);
            all.insert(all.end(), (size_t)cfg.nNeutronA, cfg.neutronA);
            all.insert(all.end(), (size_t)cfg.nProtonB, cfg.protonB);
            all.insert(all.end(), (size_t)cfg.nNeutronB, cfg.neutronB);
        }
        static const char *const kinds[] = {"ws", "r2ws", "gauss", "r2gauss"};
        for (size_t i = 0; i < all.size();) {
            size_t j = i;
            while (j < all.size() && all[j].kind == all[i].kind && all[j].p0 == all[i].p0 && all[j].p1 == all[i].p1) ++j;
            std::printf("density %zu..%zu %s:%.9g:%.9g\n", i, j - 1, all[i].kind >= 0 && all[i].kind < 4 ? kinds[all[i].kind] : "?",
                        all[i].p0, all[i].p1);
            i = j;
        }
        return 0;
    }
    if (mcmc_device_count() <= 0) {
        std::fprintf(stderr, "xsection: no HIP device (%s); this build has no CPU path\n", mcmc_last_error());
        return 3;
    }

    const int nNucleonsA = cfg.nNeutronA + cfg.nProtonA, nNucleonsB = cfg.nNeutronB + cfg.nProtonB;
    const int nNucleons = nNucleonsA + nNucleonsB;
    const double totalThreads = (double)cfg.totalBlocks * cfg.threadsPerBlock; // main.cu:89-92
    const double samplesPerRun = cfg.samplesPerThread * totalThreads;
    const double totalSamples = samplesPerRun * cfg.totalRuns;
    std::printf("%s\nnucleus A: %d nucleons, nucleus B: %d nucleons\n", mcmc_version(), nNucleonsA, nNucleonsB);
    std::printf("per integral: %.0f threads x %d samples x %d runs = %.0f samples (log10 %.2f); S_AB scan: %d impact "
                "parameters, %.3g samples (log10 %.2f)\n",
                totalThreads, cfg.samplesPerThread, cfg.totalRuns, totalSamples, std::log10(totalSamples), cfg.ndbs,
                totalSamples * cfg.ndbs, std::log10(totalSamples * cfg.ndbs));

    if (autoStep) { // the stepsize study for nucleus A's neutron density; widths from 0.25 to 2 radii
        const mcmc_target &t = cfg.neutronA;
        const double scale = t.kind == MCMC_TARGET_WS || t.kind == MCMC_TARGET_R2WS ? t.p0 + 2.0 * t.p1 : std::sqrt((double)t.p0);
        double best = 0.0;
        if (mcmc_tune_stepsize(&t, 0.25 * scale, 2.0 * scale, 15, 1 << 16, 2000, 2000, cfg.seed, cfg.initSample, nullptr, nullptr,
                               nullptr, &best))
            die("mcmc_tune_stepsize");
        cfg.stepsize = (float)best;
        std::printf("proposal width tuned on the device: %.3f (reference constant: 2.2)\n", best);
    }

    // one GPU: the Layer B handle; devices=...: the same calls on the sharded job (Layer C)
    mcmc_handle h = nullptr;
    mcmc_multi m = nullptr;
    if (devices.empty()) {
        if (mcmc_create(&cfg, &h)) die("mcmc_create");
    } else {
        if (mcmc_multi_create(&cfg, devices.data(), (int)devices.size(), reduce, &m)) die("mcmc_multi_create");
        int used = 0;
        std::vector<uint64_t> cnt(devices.size());
        mcmc_multi_info(m, nullptr, &used, nullptr, cnt.data());
        std::printf("%zu shards (events per shard: %llu ... %llu), statistics reduced by %s\n", devices.size(),
                    (unsigned long long)cnt.front(), (unsigned long long)cnt.back(), used == MCMC_REDUCE_RCCL ? "RCCL all-reduce" : "host sum");
    }
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };

    if (randomStart && (m ? mcmc_multi_init_from_pdf(m) : mcmc_init_from_pdf(h))) die("mcmc_init_from_pdf"); // main.cu:144-148
    auto t0 = now();
    if (m ? mcmc_multi_warmup(m, cfg.stepsize, cfg.samplesForMix) : mcmc_warmup(h, cfg.stepsize, cfg.samplesForMix)) die("mcmc_warmup"); // main.cu:149
    std::printf("Metropolis chain warm-up (%d sweeps): %.1f ms\n", cfg.samplesForMix, ms(t0, now()));

    if (integrateRMS) { // main.cu:157-286
        std::vector<float> rms(nNucleons), err(nNucleons);
        t0 = now();
        if (m ? mcmc_multi_run_rms(m, rms.data(), err.data()) : mcmc_run_rms(h, rms.data(), err.data())) die("mcmc_run_rms");
        const double t = ms(t0, now());
        std::printf("RMS results (%.1f ms, %.3g proposals/s)\n", t, totalSamples * nNucleons / (t * 1e-3));
        for (int n = 0; n < nNucleons; ++n) std::printf("Nucleon %d has RMS = %g +/- %g\n", n, rms[n], err[n]); // main.cu:264
        if (!mcmc_host::write_rms_csv(rms, err, nNucleonsA, outDir + outFile + ".RMS.csv")) {
            std::fprintf(stderr, "xsection: cannot write %s%s.RMS.csv\n", outDir.c_str(), outFile.c_str());
            return 1;
        }
    }

    if (integrateSAB && inelOnly) { // the b loop of main.cu:289-413 for the black-disk inelastic probability alone
        const int n = cfg.ndbs;
        std::vector<float> b(n);
        std::vector<double> pin(n), perr(n);
        const float dmax2 = cfg.sigmaNN / 3.14159265358979f;
        t0 = now();
        if (m ? mcmc_multi_run_inel_scan(m, b.data(), pin.data(), perr.data(), dmax2)
              : mcmc_run_inel_scan(h, b.data(), pin.data(), perr.data(), dmax2))
            die("mcmc_run_inel_scan");
        const double t = ms(t0, now());
        std::printf("inelastic scan: %d impact parameters in %.1f ms (%.3g event samples/s)\n", n, t, totalSamples * n / (t * 1e-3));
        if (!mcmc_host::write_inel_csv(b, pin, perr, outDir + outFile + ".inel.csv")) {
            std::fprintf(stderr, "xsection: cannot write %s%s.inel.csv\n", outDir.c_str(), outFile.c_str());
            return 1;
        }
        if (n >= 2) { // sigma_inel = 2 pi int b p_inel(b) db (trapezoid), x10: fm^2 -> mb; errors added in quadrature
            double s = 0, v = 0;
            for (int i = 0; i < n; ++i) {
                const double w = 0.5 * ((i > 0 ? b[i] - b[i - 1] : 0.0) + (i + 1 < n ? b[i + 1] - b[i] : 0.0)) * b[i];
                s += w * pin[i];
                v += w * w * perr[i] * perr[i];
            }
            const double k = 10 * 2 * 3.14159265358979323846;
            std::printf("sigma_inel (black disk) = %.3f +/- %.3f mb\n", k * s, k * std::sqrt(v));
        }
    } else if (integrateSAB) { // main.cu:289-413
        const int n = cfg.ndbs;
        std::vector<float> b(n), S(n), E(n);
        std::vector<double> pin(n);
        const float dmax2 = cfg.sigmaNN / 3.14159265358979f; // black disk: sigma_NN = pi d_max^2
        t0 = now();
        if (m ? mcmc_multi_run_sab_scan(m, b.data(), S.data(), E.data(), blackDisk ? pin.data() : nullptr, dmax2)
              : mcmc_run_sab_scan(h, b.data(), S.data(), E.data(), blackDisk ? pin.data() : nullptr, dmax2))
            die("mcmc_run_sab_scan");
        const double t = ms(t0, now());
        std::printf("S_AB scan: %d impact parameters in %.1f ms (%.3g event samples/s, %.3g nominal pair terms/s)\n", n, t,
                    totalSamples * n / (t * 1e-3), totalSamples * n * nNucleonsA * nNucleonsB / (t * 1e-3));
        if (!mcmc_host::write_sab_csv(b, S, E, blackDisk ? &pin : nullptr, outDir + outFile + ".S_AB.csv")) {
            std::fprintf(stderr, "xsection: cannot write %s%s.S_AB.csv\n", outDir.c_str(), outFile.c_str());
            return 1;
        }
        if (n >= 2) { // XSectionAnalysis.py:10-28
            std::vector<double> bd(b.begin(), b.end()), Sd(S.begin(), S.end()), Ed(E.begin(), E.end());
            double cs[6];
            if (mcmc_cross_sections(bd.data(), Sd.data(), Ed.data(), n, cs)) die("mcmc_cross_sections");
            std::printf("sigma_tot = %.3f +/- %.3f mb\nsigma_el  = %.3f +/- %.3f mb\nsigma_rxn = %.3f +/- %.3f mb\n", cs[0], cs[1],
                        cs[2], cs[3], cs[4], cs[5]);
            if (blackDisk) { // sigma_inel = 2 pi int b p_inel(b) db (trapezoid), x10: fm^2 -> mb
                double s = 0;
                for (int i = 0; i + 1 < n; ++i) s += 0.5 * (bd[i] * pin[i] + bd[i + 1] * pin[i + 1]) * (bd[i + 1] - bd[i]);
                std::printf("sigma_inel (black disk) = %.3f mb\n", 10 * 2 * 3.14159265358979323846 * s);
            }
        }
    }
    if (m ? mcmc_multi_destroy(m) : mcmc_destroy(h)) die("mcmc_destroy"); // main.cu:419-422
    return 0;
}
