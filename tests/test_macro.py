"""`xsection macro=<file>`: macro files written by the reference's generator (IntegrationSettings.GenerateCode,
xsection/tools/MCMCMacro.py:285-372 -- the reference compiles one executable per macro, main.cu:22-37) read at run
time.  Fixtures: tests/golden/macro_*.mac are OUTPUTS of that generator (tests/golden/make_macro_fixtures.py, run in the
build container) and macro_expected.json holds the settings that produced them.
CPU part: `printConfig=1` (no device needed) must recover every setting and every nucleon's density, in pdfArr order.
GPU part: the per-nucleon densities reach the device in that order (each nucleon's RMS against quadrature), and the
sharded job reads the same macro."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XSECTION = os.path.join(ROOT, "mcmc_amd", "host", "xsection")
GOLDEN = os.path.join(ROOT, "tests", "golden")
EXPECTED = json.load(open(os.path.join(GOLDEN, "macro_expected.json")))
KIND = {"WoodsSaxon": "ws", "R2WoodsSaxon": "r2ws", "Gaussian": "gauss", "R2Gaussian": "r2gauss"}


def print_config(*args):
    r = subprocess.run([XSECTION, *args, "printConfig=1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    cfg, dens = {}, []
    for line in r.stdout.splitlines():
        w = line.split()
        if w[0] == "density":
            lo, hi = (int(x) for x in w[1].split(".."))
            kind, p0, p1 = w[2].split(":")
            dens += [(kind, np.float32(p0), np.float32(p1))] * (hi - lo + 1)
        elif not line.startswith("macro:"):
            cfg[w[0]] = w[1] if len(w) > 1 else ""
    return cfg, dens, r.stdout


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_macro_settings_and_per_nucleon_densities_are_recovered(name):
    exp = EXPECTED[name]
    cfg, dens, out = print_config(f"macro={os.path.join(GOLDEN, name)}")
    for key in ("nNeutronA", "nProtonA", "nNeutronB", "nProtonB", "totalBlocks", "threadsPerBlock", "samplesPerThread",
                "totalRuns", "samplesForMix", "ndbs"):
        assert int(cfg[key]) == exp[key], key
    assert np.float32(cfg["b0"]) == np.float32(exp["b0"]) and np.float32(cfg["db"]) == np.float32(exp["db"])
    assert int(cfg["integrateRMS"]) == int(exp["integrateRMS"]) and int(cfg["integrateSAB"]) == int(exp["integrateSAB"])
    assert cfg["outDir"] == exp["outDir"]
    assert cfg["outFile"] == exp["outFile"][:-len(".root")]       # the CSV tables take the ROOT file's stem
    if exp["seed"] is None:                                       # seeded from the clock there: the fixed default, and said so
        assert int(cfg["seed"]) == 314159265 and "seed taken from the clock" in out
    else:
        assert int(cfg["seed"]) == exp["seed"]
    assert len(dens) == len(exp["nucleons"]) == exp["nNeutronA"] + exp["nProtonA"] + exp["nNeutronB"] + exp["nProtonB"]
    for n, (got, want) in enumerate(zip(dens, exp["nucleons"])):
        p = list(want["params"].values())                         # (c, a) or (variance), in the callable's argument order
        assert got[0] == KIND[want["func"]], n
        assert got[1] == np.float32(p[0]) and got[2] == np.float32(p[1] if len(p) > 1 else 0.0), n


def test_arguments_after_the_macro_override_it():
    mac = os.path.join(GOLDEN, "macro_c12_per_nucleon.mac")
    cfg, dens, _ = print_config(f"macro={mac}", "ndbs=5", "seed=7", "outFile=mine")
    assert cfg["ndbs"] == "5" and cfg["seed"] == "7" and cfg["outFile"] == "mine"
    assert len({d for d in dens}) == 3                            # the macro's three distinct densities are kept
    cfg, dens, _ = print_config(f"macro={mac}", "pdfA=r2ws:3:0.5")  # a density argument replaces the per-nucleon list
    assert len(dens) == 24 and all(d == ("r2ws", np.float32(3), np.float32(0.5)) for d in dens[:12])
    assert all(d[0] in ("ws", "r2ws") for d in dens[12:])
    cfg, dens, _ = print_config("isotopeA=HE-4", f"macro={mac}")    # ... and the macro replaces what came before it
    assert cfg["nProtonA"] == "6" and len(dens) == 24


def test_hand_written_block_of_the_reference_main(tmp_path):
    """The constant block of main.cu:38-82 in its own syntax (ConstArray<nNucleonsA>(alias), shifts, 2.f literals) --
    written here, not copied: two densities, 3 + 5 nucleons."""
    mac = tmp_path / "own.mac"
    mac.write_text("""
__device__ __host__ float small(float r) {
    const float c = 2.f;   // radius
    const float a = 0.54f;
    return R2WoodsSaxon(r, c, a);
}
__device__ __host__ float wide(float r) { return Gaussian(r, 1.5f * 2); }
    const int nNeutronA = 1;
    const int nProtonA = 2;
    const int nNeutronB = 3;
    const int nProtonB = 2;
    const int nNucleonsA = nNeutronA + nProtonA;
    const int nNucleonsB = nNeutronB + nProtonB;
    constexpr static PDF nucleusAPdf = small;
    constexpr static PDF nucleusBPdf = wide;
    constexpr static auto pdfArrA = TemplateBuilder::ConstArray<nNucleonsA>(nucleusAPdf);
    constexpr static auto pdfArrB = TemplateBuilder::ConstArray<nNucleonsB>(nucleusBPdf);
    constexpr static auto pdfArr  = TemplateBuilder::ConcatArray(pdfArrA, pdfArrB);
    const int totalBlocks = 1 << 6; // 64
    const int threadsPerBlock = 1 << 7;
    const float db = 0.1;
    const uint64_t seed = 18446744073709551557;
    std::string outFile = "own.root";
    if (argc == 2) {
        outFile = argv[1];
    }
    const bool integrateRMS = false;
    const int ndbs = (1 << 3) + 2 * 3 - 4 / 2;
""")
    cfg, dens, _ = print_config(f"macro={mac}")
    assert (cfg["totalBlocks"], cfg["threadsPerBlock"], cfg["integrateRMS"], cfg["outFile"]) == ("64", "128", "0", "own")
    assert cfg["ndbs"] == "12"
    assert cfg["seed"] == "18446744073709551557"                  # all 64 bits of the seed
    assert dens == [("r2ws", np.float32(2.0), np.float32(0.54))] * 3 + [("gauss", np.float32(3.0), np.float32(0.0))] * 5


@pytest.mark.parametrize("text, message", [
    ("__device__ float f(float r) { return Lorentzian(r, 1.0); }\nconst int nProtonA = 1;", "unknown density"),
    ("__device__ float f(float r) { float c = 1.0; return R2WoodsSaxon(r, c); }", "wrong number of parameters"),
    ("constexpr static auto pdfArr = TemplateBuilder::ConstArray<2>(nothing);", "not a density function"),
    ("__device__ float f(float r) { return Gaussian(r, 1.0); }\nconst int nProtonA = 1;\nconst int nNeutronA = 0;\n"
     "const int nProtonB = 1;\nconst int nNeutronB = 0;\nconstexpr static auto pdfArr = TemplateBuilder::ConstArray<3>(f);",
     "pdfArr has 3 densities for 1 + 1 nucleons"),
])
def test_unreadable_macros_are_refused_with_the_reason(tmp_path, text, message):
    mac = tmp_path / "bad.mac"
    mac.write_text(text)
    r = subprocess.run([XSECTION, f"macro={mac}", "printConfig=1"], capture_output=True, text=True)
    assert r.returncode == 2 and message in r.stderr, r.stderr
    r = subprocess.run([XSECTION, f"macro={tmp_path / 'missing.mac'}", "printConfig=1"], capture_output=True, text=True)
    assert r.returncode == 2 and "cannot open" in r.stderr


def _rms_quadrature(kind, p0, p1):
    from scipy import integrate
    f = {"ws": lambda r: 1.0 / (1.0 + np.exp((r - p0) / p1)), "r2ws": lambda r: r * r / (1.0 + np.exp((r - p0) / p1)),
         "gauss": lambda r: np.exp(-r * r / (2 * p0)), "r2gauss": lambda r: r * r * np.exp(-r * r / (2 * p0))}[kind]
    hi = p0 + 60 * p1 if kind in ("ws", "r2ws") else 14 * np.sqrt(p0)
    z = integrate.quad(f, 0, hi, epsabs=0, epsrel=1e-12)[0]
    return np.sqrt(integrate.quad(lambda r: r * r * f(r), 0, hi, epsabs=0, epsrel=1e-12)[0] / z)


@pytest.mark.gpu
def test_macro_per_nucleon_densities_reach_the_device_in_order(dev, tmp_path):
    """C-12 + C-12 with neutron 3 of A on R2Gaussian(2.5) and proton 0 of B on WoodsSaxon(2.4, 0.45): every nucleon's RMS
    (the RMS table of main.cu:253-258) agrees with the quadrature value of ITS density; one job on device 0 and the
    same job as two shards (devices=0,0) write identical tables."""
    mac = os.path.join(GOLDEN, "macro_c12_per_nucleon.mac")
    r = subprocess.run([XSECTION, f"macro={mac}", "totalBlocks=64", "totalRuns=8", "ndbs=4", "db=1.5", f"outDir={tmp_path}",
                        "outFile=one"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    rms = np.genfromtxt(tmp_path / "one.RMS.csv", delimiter=",", names=True)
    assert len(rms) == 24
    want = np.full(24, _rms_quadrature("r2ws", 2.1, 0.5))
    want[9] = _rms_quadrature("r2gauss", 2.5, 0.0)     # protons A 0..5, neutrons A 6..11: neutron 3 is nucleon 9
    want[12] = _rms_quadrature("ws", 2.4, 0.45)        # protons B start at 12
    assert want[9] == pytest.approx(np.sqrt(7.5), rel=1e-9) and abs(want[12] - want[0]) > 0.3
    assert np.all(np.abs(rms["RMS"] - want) < np.maximum(5 * rms["RMS_stderr"], 3e-3)), (rms["RMS"], want)
    r2 = subprocess.run([XSECTION, f"macro={mac}", "totalBlocks=64", "totalRuns=8", "ndbs=4", "db=1.5", f"outDir={tmp_path}",
                         "outFile=two", "devices=0,0", "reduce=host"], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr + r2.stdout
    for table in ("RMS", "S_AB"):
        a = np.genfromtxt(tmp_path / f"one.{table}.csv", delimiter=",", names=True)
        b = np.genfromtxt(tmp_path / f"two.{table}.csv", delimiter=",", names=True)
        for col in a.dtype.names:
            np.testing.assert_allclose(a[col], b[col], rtol=1e-5, atol=0)   # Layer B against Layer C: equal to the estimator's rounding (and one printed digit)
