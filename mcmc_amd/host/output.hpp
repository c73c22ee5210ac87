// output.hpp -- result files of the two drivers, in the reference's layouts.
//   statistics.txt / samples_<k>.txt : the text formats of stepsize/include/fileio.hpp:8-43
//     (one sample per line; header "# Stepsize Acceptance k=1 ... k=K", rows of
//     space-separated values with a trailing space).
//   <out>.RMS.csv / <out>.S_AB.csv   : the TTree columns of
//     xsection/include/IntegrationOutputHandler.cuh:83-87,109-112 (Nucleus, Nucleon, RMS,
//     RMS_stderr; b, S_AB, S_AB_stderr) as CSV -- ROOT is not part of this build.
#pragma once
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

namespace mcmc_host {

inline bool write_samples(const std::vector<double> &samples, const std::string &path)
{
    std::ofstream f(path);
    if (!f) return false;
    for (double s : samples) f << s << "\n";
    return (bool)f;
}

// rows[i] = {stepsize, acceptance, rho_1 ... rho_K}
inline bool write_statistics(const std::vector<std::vector<double>> &rows, const std::string &path)
{
    std::ofstream f(path);
    if (!f || rows.empty()) return false;
    f << "# Stepsize Acceptance";
    for (size_t k = 1; k + 1 < rows[0].size(); ++k) f << " k=" << k;
    f << "\n";
    for (const auto &r : rows) {
        for (double v : r) f << v << " ";
        f << "\n";
    }
    return (bool)f;
}

inline bool write_rms_csv(const std::vector<float> &rms, const std::vector<float> &err, int nNucleonsA, const std::string &path)
{
    std::ofstream f(path);
    if (!f) return false;
    f << "Nucleus,Nucleon,RMS,RMS_stderr\n";
    for (size_t i = 0; i < rms.size(); ++i) {
        const bool inA = (int)i < nNucleonsA; // IntegrationOutputHandler.cuh:90-97: 1-based, nucleus 1 then 2
        f << (inA ? 1 : 2) << "," << (inA ? i + 1 : i + 1 - nNucleonsA) << "," << rms[i] << "," << err[i] << "\n";
    }
    return (bool)f;
}

inline bool write_sab_csv(const std::vector<float> &b, const std::vector<float> &S, const std::vector<float> &E,
                          const std::vector<double> *p_inel, const std::string &path)
{
    std::ofstream f(path);
    if (!f) return false;
    f << "b,S_AB,S_AB_stderr" << (p_inel ? ",p_inel" : "") << "\n";
    char line[160];
    for (size_t i = 0; i < b.size(); ++i) {
        if (p_inel) std::snprintf(line, sizeof line, "%.9g,%.9g,%.9g,%.9g\n", b[i], S[i], E[i], (*p_inel)[i]);
        else std::snprintf(line, sizeof line, "%.9g,%.9g,%.9g\n", b[i], S[i], E[i]);
        f << line;
    }
    return (bool)f;
}

// Black-disk scan: columns b, p_inel, p_inel_stderr (no counterpart in the reference's TTree).
inline bool write_inel_csv(const std::vector<float> &b, const std::vector<double> &p, const std::vector<double> &e,
                           const std::string &path)
{
    std::ofstream f(path);
    if (!f) return false;
    f << "b,p_inel,p_inel_stderr\n";
    char line[160];
    for (size_t i = 0; i < b.size(); ++i) {
        std::snprintf(line, sizeof line, "%.9g,%.12g,%.6g\n", b[i], p[i], e[i]);
        f << line;
    }
    return (bool)f;
}

} // namespace mcmc_host
