// macro.hpp -- reads, at RUN time, a macro file written by the reference's generator
// (IntegrationSettings.GenerateCode, xsection/tools/MCMCMacro.py:285-372).  The reference compiles one executable
// per macro: main.cu:22-37 #includes the file twice, once for the density functions (EXTRACT_MACRO_FUNCTIONS) and once
// for the constant block that replaces main.cu:38-82.  Here the same file is data: `xsection macro=<file>`.
//
// What is understood (the generator's whole output, and the hand-written block of main.cu:38-82):
//   __device__ __host__ float NAME(float r) { float c = 2.0; float a = 0.54; return R2WoodsSaxon(r, c, a); }
//       -> a density: one of the four callables of Distributions.cuh:7-28 with its parameters (MCMCMacro.py:31-49)
//   const int / const float / const bool / const uint64_t NAME = <expression>;   (+ - * / << >> and earlier names)
//   std::string NAME = "...";
//   constexpr static PDF ALIAS = NAME;                                            (main.cu:47-48)
//   constexpr static auto ARR = TemplateBuilder::ConstArray<N>(NAME);             (MCMCMacro.py:318-335, main.cu:49-50)
//   constexpr static auto ARR = TemplateBuilder::ConcatArray(ARR1, ARR2);         (MCMCMacro.py:339-343, main.cu:51)
//   constexpr static auto ARR = ARR1;                                             (MCMCMacro.py:345)
// `pdfArr` is the per-nucleon density list (protons A, neutrons A, protons B, neutrons B).  A seed taken from the
// clock (MCMCMacro.py:360) leaves the library's fixed default in place and is reported.  integrateIAB / integrateJAB
// are read and reported: the reference's main.cu has no such integrals either.
#pragma once
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "mcmc_hip.h"

namespace mcmc_host {

struct MacroSettings {
    std::map<std::string, double> numbers;       // every const int / float / bool / uint64_t the macro assigns
    std::map<std::string, std::string> strings;  // outDir, outFile
    bool seed_from_clock = false;                // seed = std::chrono::...: not reproducible, the library default stays
    bool has_seed = false;                       // seed = <integer literal> (kept apart: 64 bits do not fit a double)
    uint64_t seed = 0;
    std::vector<mcmc_target> nucleons;           // pdfArr, one density per nucleon (empty: the macro defines none)
    std::vector<std::string> notes;              // what was skipped, for the driver to print
};

namespace macro_detail {

inline std::string trim(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

inline bool is_ident(const std::string &s)
{
    if (s.empty() || !(std::isalpha((unsigned char)s[0]) || s[0] == '_')) return false;
    for (char c : s)
        if (!(std::isalnum((unsigned char)c) || c == '_')) return false;
    return true;
}

// <expression> of integer / float literals (2.f, 0.54f, 1 << 8), true / false and names assigned earlier
struct Expr {
    const std::string &s;
    const std::map<std::string, double> &names;
    size_t i = 0;
    bool ok = true;
    void ws()
    {
        while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
    }
    double primary()
    {
        ws();
        if (i >= s.size()) return ok = false, 0.0;
        if (s[i] == '(') {
            ++i;
            const double v = shift();
            ws();
            if (i < s.size() && s[i] == ')') ++i;
            else ok = false;
            return v;
        }
        if (std::isdigit((unsigned char)s[i]) || s[i] == '.') {
            char *end = nullptr;
            const double v = std::strtod(s.c_str() + i, &end);
            i = (size_t)(end - s.c_str());
            while (i < s.size() && (s[i] == 'f' || s[i] == 'F' || s[i] == 'u' || s[i] == 'U' || s[i] == 'l' || s[i] == 'L')) ++i;
            return v;
        }
        size_t j = i;
        while (j < s.size() && (std::isalnum((unsigned char)s[j]) || s[j] == '_')) ++j;
        const std::string id = s.substr(i, j - i);
        i = j;
        if (id == "true") return 1.0;
        if (id == "false") return 0.0;
        const auto it = names.find(id);
        if (id.empty() || it == names.end()) return ok = false, 0.0;
        return it->second;
    }
    double unary()
    {
        ws();
        if (i < s.size() && s[i] == '-') return ++i, -unary();
        if (i < s.size() && s[i] == '+') return ++i, unary();
        return primary();
    }
    double mul()
    {
        double v = unary();
        for (;;) {
            ws();
            if (i < s.size() && s[i] == '*') ++i, v *= unary();
            else if (i < s.size() && s[i] == '/') ++i, v /= unary();
            else return v;
        }
    }
    double add()
    {
        double v = mul();
        for (;;) {
            ws();
            if (i < s.size() && s[i] == '+') ++i, v += mul();
            else if (i < s.size() && s[i] == '-') ++i, v -= mul();
            else return v;
        }
    }
    double shift()
    {
        double v = add();
        for (;;) {
            ws();
            const bool left = i + 1 < s.size() && s[i] == '<' && s[i + 1] == '<';
            const bool right = i + 1 < s.size() && s[i] == '>' && s[i + 1] == '>';
            if (!left && !right) return v;
            i += 2;
            const double n = add();
            if (!(n >= 0.0 && n <= 62.0) || !(v > -9.0e18 && v < 9.0e18)) return ok = false, 0.0;
            v = left ? (double)((long long)v << (int)n) : (double)((long long)v >> (int)n);
        }
    }
};

inline bool eval(const std::string &text, const std::map<std::string, double> &names, double &out)
{
    Expr e{text, names};
    out = e.shift();
    e.ws();
    return e.ok && e.i == text.size();
}

// a declaration "qualifiers type NAME" -> its last word, and whether `word` is among the ones before it
inline bool split_decl(const std::string &lhs, std::string &name, std::vector<std::string> &quals)
{
    std::istringstream is(lhs);
    std::vector<std::string> w;
    for (std::string t; is >> t;) w.push_back(t);
    if (w.empty() || !is_ident(w.back())) return false;
    name = w.back();
    w.pop_back();
    quals = w;
    return true;
}

inline bool has(const std::vector<std::string> &q, const char *word)
{
    for (const auto &t : q)
        if (t == word) return true;
    return false;
}

} // namespace macro_detail

// One density function of the macro -> mcmc_target.  body = the text between its braces.
inline bool macro_density(const std::string &body, mcmc_target &t, std::string &err)
{
    using namespace macro_detail;
    std::map<std::string, double> locals;
    std::istringstream is(body);
    for (std::string st; std::getline(is, st, ';');) {
        st = trim(st);
        if (st.empty()) continue;
        if (st.compare(0, 6, "return") == 0) {
            const size_t lp = st.find('('), rp = st.rfind(')');
            if (lp == std::string::npos || rp == std::string::npos || rp < lp) return err = "cannot read '" + st + "'", false;
            const std::string fn = trim(st.substr(6, lp - 6));
            std::vector<double> args;
            std::istringstream as(st.substr(lp + 1, rp - lp - 1));
            int k = 0;
            for (std::string a; std::getline(as, a, ','); ++k) {
                if (k == 0) continue; // r
                double v;
                if (!eval(trim(a), locals, v)) return err = "cannot evaluate argument '" + trim(a) + "' of " + fn, false;
                args.push_back(v);
            }
            // Distributions.cuh:7-28
            const bool two = fn == "WoodsSaxon" || fn == "R2WoodsSaxon", one = fn == "Gaussian" || fn == "R2Gaussian";
            if (!two && !one) return err = "unknown density '" + fn + "' (WoodsSaxon, R2WoodsSaxon, Gaussian, R2Gaussian)", false;
            if (args.size() != (two ? 2u : 1u)) return err = fn + ": wrong number of parameters", false;
            t.kind = fn == "WoodsSaxon" ? MCMC_TARGET_WS : fn == "R2WoodsSaxon" ? MCMC_TARGET_R2WS
                     : fn == "Gaussian" ? MCMC_TARGET_GAUSS : MCMC_TARGET_R2GAUSS;
            t.p0 = (float)args[0];
            t.p1 = two ? (float)args[1] : 0.0f;
            return true;
        }
        const size_t eq = st.find('=');
        if (eq == std::string::npos) continue;
        std::string name;
        std::vector<std::string> quals;
        double v;
        if (!split_decl(st.substr(0, eq), name, quals) || !eval(trim(st.substr(eq + 1)), locals, v))
            return err = "cannot read '" + st + "'", false;
        locals[name] = v;
    }
    return err = "a density function without a return statement", false;
}

inline bool read_macro(const std::string &path, MacroSettings &out, std::string &err)
{
    using namespace macro_detail;
    std::ifstream f(path);
    if (!f) return err = "cannot open " + path, false;
    std::string text, line;
    while (std::getline(f, line)) {
        const size_t c = line.find("//");
        if (c != std::string::npos) line.erase(c);
        const std::string t = trim(line);
        if (!t.empty() && t[0] == '#') continue; // #ifdef EXTRACT_MACRO_FUNCTIONS / #else / #endif
        text += line;
        text += '\n';
    }

    // density functions: "... float NAME(float r) { body }" -- cut out of the text as they are read
    std::map<std::string, mcmc_target> dens;
    for (size_t lb; (lb = text.find('{')) != std::string::npos;) {
        const size_t rb = text.find('}', lb);
        if (rb == std::string::npos) return err = "unbalanced braces", false;
        size_t start = text.rfind(';', lb);
        const size_t prev_rb = text.rfind('}', lb);
        start = start == std::string::npos ? 0 : start + 1;
        if (prev_rb != std::string::npos && prev_rb + 1 > start) start = prev_rb + 1;
        const std::string head = text.substr(start, lb - start);
        const size_t lp = head.find('(');
        std::string name;
        std::vector<std::string> quals;
        if (lp == std::string::npos || !split_decl(head.substr(0, lp), name, quals))
            return err = "cannot read the block that starts '" + trim(head) + "'", false;
        if (name == "if" || name == "for" || name == "while" || name == "else" || name == "switch") {
            // control flow of the hand-written block (main.cu:76-79: the lone argument as output file): not a setting
            out.notes.push_back("skipped: " + trim(head) + " { ... }");
            text.erase(start, rb + 1 - start);
            continue;
        }
        mcmc_target t{};
        std::string why;
        if (!macro_density(text.substr(lb + 1, rb - lb - 1), t, why)) return err = name + ": " + why, false;
        dens[name] = t;
        text.erase(start, rb + 1 - start);
    }

    std::map<std::string, std::vector<std::string>> arrays; // array name -> density names
    std::map<std::string, std::string> alias;               // constexpr static PDF a = f
    auto resolve = [&](std::string n) {
        for (int hop = 0; hop < 8 && alias.count(n); ++hop) n = alias[n];
        return n;
    };
    std::istringstream is(text);
    for (std::string st; std::getline(is, st, ';');) {
        st = trim(st);
        if (st.empty()) continue;
        const size_t eq = st.find('=');
        std::string name;
        std::vector<std::string> quals;
        if (eq == std::string::npos || !split_decl(st.substr(0, eq), name, quals)) {
            out.notes.push_back("skipped: " + st);
            continue;
        }
        const std::string rhs = trim(st.substr(eq + 1));
        if (has(quals, "std::string") || has(quals, "string")) {
            const size_t a = rhs.find('"'), b = rhs.rfind('"');
            if (a == std::string::npos || b <= a) return err = "cannot read the string '" + st + "'", false;
            out.strings[name] = rhs.substr(a + 1, b - a - 1);
        } else if (has(quals, "PDF")) {
            alias[name] = rhs;
        } else if (has(quals, "auto")) {
            const size_t lp = rhs.find('('), rp = rhs.rfind(')');
            if (rhs.find("ConstArray") != std::string::npos && lp != std::string::npos && rp != std::string::npos) {
                const size_t la = rhs.find('<'), ra = rhs.find('>');
                double cnt;
                if (la == std::string::npos || ra == std::string::npos || ra > lp || !eval(trim(rhs.substr(la + 1, ra - la - 1)), out.numbers, cnt) ||
                    cnt < 0 || cnt > 100000)
                    return err = "cannot read the element count of '" + st + "'", false;
                const std::string fn = resolve(trim(rhs.substr(lp + 1, rp - lp - 1)));
                if (!dens.count(fn)) return err = name + ": '" + fn + "' is not a density function of this macro", false;
                arrays[name].assign((size_t)cnt, fn);
            } else if (rhs.find("ConcatArray") != std::string::npos && lp != std::string::npos && rp != std::string::npos) {
                std::istringstream as(rhs.substr(lp + 1, rp - lp - 1));
                std::vector<std::string> all;
                for (std::string a; std::getline(as, a, ',');) {
                    a = trim(a);
                    if (!arrays.count(a)) return err = name + ": '" + a + "' is not an array defined before it", false;
                    all.insert(all.end(), arrays[a].begin(), arrays[a].end());
                }
                arrays[name] = all;
            } else if (arrays.count(rhs)) {
                arrays[name] = arrays[rhs];
            } else {
                out.notes.push_back("skipped: " + st);
            }
        } else {
            double v;
            if (name == "seed" && !rhs.empty() && rhs.find_first_not_of("0123456789uUlL") == std::string::npos) {
                out.has_seed = true;
                out.seed = std::strtoull(rhs.c_str(), nullptr, 10);
            } else if (eval(rhs, out.numbers, v)) {
                out.numbers[name] = v;
                if (name == "seed" && v >= 0.0 && v < 1.8e19) out.has_seed = true, out.seed = (uint64_t)v;
            } else if (name == "seed") {
                out.seed_from_clock = true; // std::chrono::steady_clock::now()... (MCMCMacro.py:360, main.cu:69)
            } else {
                out.notes.push_back("skipped: " + st);
            }
        }
    }
    if (arrays.count("pdfArr"))
        for (const auto &fn : arrays["pdfArr"]) out.nucleons.push_back(dens[fn]);
    return true;
}

} // namespace mcmc_host
