"""include/mcmc_distributions.h -- the reference's target-density callables under their own names
(xsection/include/Distributions.cuh:7-28, stepsize/include/metropolis.hpp:9-17), part of the drop-in boundary
(SURVEY 8b).  Compiled with g++ on the host and with hipcc for the device; values against the reference's own
outputs (tests/golden/stepsize_a1.npz: ref_R2WoodsSaxon_*, ref_R2Gaussian_1, produced by the reference's header)
and the oracle's FP32 densities."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "stepsize_a1.npz")

HOST_PROBE = r'''
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "mcmc_distributions.h"
static unsigned long long b64(double d) { unsigned long long u; std::memcpy(&u, &d, 8); return u; }
static unsigned b32(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
int main(int argc, char **argv)
{
    // argv: radii as decimal strings with 17 significant digits (exact round trip)
    for (int i = 1; i < argc; ++i) {
        const double r = std::strtod(argv[i], nullptr);
        const float rf = (float)r;
        std::printf("%016llx %016llx %016llx %08x %08x %08x %08x\n", b64(R2WoodsSaxon(r, 2.0, 0.54)),
                    b64(R2WoodsSaxon(r, 6.62, 0.546)), b64(R2Gaussian(r, 1.0)), b32(WoodsSaxon(rf, 2.0f, 0.54f)),
                    b32(R2WoodsSaxon(rf, 2.0f, 0.54f)), b32(Gaussian(rf, 1.5f)), b32(R2Gaussian(rf, 1.0f)));
    }
    return 0;
}
'''


def _f32(u):
    return np.array([u], dtype=np.uint32).view(np.float32)[0]


def test_header_on_the_host_matches_reference_values_and_oracle(tmp_path, oracle):
    g = np.load(GOLD)
    r = g["pdf_r"]
    src = tmp_path / "probe.cpp"
    src.write_text(HOST_PROBE)
    exe = tmp_path / "probe"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                    str(exe)], check=True)
    out = subprocess.run([str(exe)] + [f"{x:.17g}" for x in r], check=True, capture_output=True, text=True).stdout.split("\n")
    import oracle_lib as O
    for i, x in enumerate(r):
        w = out[i].split()
        d = [np.array([int(v, 16)], dtype=np.uint64).view(np.float64)[0] for v in w[:3]]
        # FP64 overloads == the reference's own R2WoodsSaxon / R2Gaussian (metropolis.hpp:9-17), bit for bit
        assert d[0] == g["ref_R2WoodsSaxon_2_0p54"][i]
        assert d[1] == g["ref_R2WoodsSaxon_6p62_0p546"][i]
        assert d[2] == g["ref_R2Gaussian_1"][i]
        # FP32 callables == the oracle's restatement of Distributions.cuh:7-28, bit for bit (both glibc expf)
        xf = float(np.float32(x))
        f = [_f32(int(v, 16)) for v in w[3:]]
        assert f[0] == np.float32(oracle.target_f32(O.WS, 2.0, 0.54, xf))
        assert f[1] == np.float32(oracle.target_f32(O.R2WS, 2.0, 0.54, xf))
        assert f[2] == np.float32(oracle.target_f32(O.GAUSS, 1.5, 0.0, xf))
        assert f[3] == np.float32(oracle.target_f32(O.R2GAUSS, 1.0, 0.0, xf))
        # and the float forms follow the double ones
        assert f[1] == pytest.approx(g["ref_R2WoodsSaxon_2_0p54"][i], rel=3e-6, abs=1e-30)


def test_header_compiles_for_the_device(tmp_path):
    """hipcc --offload-arch=gfx950: the same inlines are __host__ __device__ (no GPU needed to compile)."""
    obj = tmp_path / "dist_probe.o"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    "-c", os.path.join(ROOT, "tools", "dist_probe.hip"), "-o", str(obj)], check=True)
    assert obj.stat().st_size > 0


@pytest.mark.gpu
def test_header_on_the_device_matches_oracle(oracle):
    """tools/bin/dist_probe (built by `make probes`): the callables evaluated in a kernel on the MI355X against the
    oracle's FP32 densities -- OCML expf vs glibc expf, so equal to a few ulp, exactly 0 for r < 0 and in the
    expf-overflow tail; the host side of the same translation unit is bit-equal to the oracle."""
    import oracle_lib as O
    exe = os.path.join(ROOT, "tools", "bin", "dist_probe")
    assert os.path.exists(exe), "tools/bin/dist_probe missing: run `make probes`"
    rows = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout.split("\n")
    spec = {"WoodsSaxon_2_0.54": (O.WS, 2.0, 0.54), "R2WoodsSaxon_2_0.54": (O.R2WS, 2.0, 0.54), "Gaussian_1.5": (O.GAUSS, 1.5, 0.0),
            "R2Gaussian_1": (O.R2GAUSS, 1.0, 0.0), "R2WoodsSaxon_6.62_0.546": (O.R2WS, 6.62, 0.546)}
    n = 0
    for row in rows:
        if not row.strip():
            continue
        name, rb, db, hb = row.split()
        r, dv, hv = _f32(int(rb, 16)), _f32(int(db, 16)), _f32(int(hb, 16))
        k, p0, p1 = spec[name]
        ref = np.float32(oracle.target_f32(k, p0, p1, float(r)))
        assert hv == ref, (name, r)
        if r < 0 or ref == 0:
            assert dv == ref
        elif ref < 1.1754944e-38:
            # below the smallest normal float the device's expf may flush to zero (denormal result)
            assert 0.0 <= float(dv) <= float(ref) * 1.001 + 1e-44, (name, r, dv, ref)
        else:
            assert abs(float(dv) - float(ref)) <= 4 * np.spacing(np.float32(abs(ref))) + 1e-44, (name, r, dv, ref)
        n += 1
    assert n == 5 * 96
