"""The C-ABI boundary without a GPU: libmcmc_hip.so loads, exports every symbol
include/mcmc_hip.h declares, the ctypes structures have the C layout, and the
argument checks that need no device reject bad input with a message."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mcmc_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcmc_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from mcmc_amd import lib as L
    declared = _declared_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L.lib, name), f"{name} declared in mcmc_hip.h but not exported"
    assert sorted(L.EXPORTS) == declared, "mcmc_amd.lib.EXPORTS is out of step with the header"
    assert L.lib.mcmc_version().decode().startswith("mcmc_hip")


def test_ctypes_structs_match_c_layout(tmp_path):
    """Compile a probe against the header with gcc and compare sizeof/offsetof with ctypes."""
    from mcmc_amd import lib as L
    probes = {
        "mcmc_target": (L.Target, ["kind", "p0", "p1"]),
        "mcmc_chain_args": (L.ChainArgs, ["n_chains", "n_steps", "target", "seed", "state", "accept_bits",
                                          "workspace_bytes"]),
        "mcmc_xs_args": (L.XsArgs, ["totalThreads", "stepsize", "seed", "injected_stride", "sigmaNN", "workspace_bytes"]),
        "mcmc_config": (L.Config, ["neutronA", "totalBlocks", "b0", "seed", "sigmaNN", "device", "event_offset"]),
        "mcmc_chain_summary": (L.ChainSummary, ["mean", "tau_int"]),
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mcmc_hip.h"', "int main(void){"]
    for cname, (_, fields) in probes.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f in fields:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines.append("return 0;}")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, (ct, fields) in probes.items():
        assert int(out[cname]) == C.sizeof(ct), cname
        for f in fields:
            assert int(out[f"{cname}.{f}"]) == getattr(ct, f).offset, f"{cname}.{f}"


def test_argument_errors_need_no_device():
    from mcmc_amd import lib as L
    lib = L.lib
    assert lib.mcmc_chain_run(None, None) == -1
    assert b"null" in lib.mcmc_last_error()
    a = L.ChainArgs()
    a.n_chains, a.n_coords = 0, 1
    assert lib.mcmc_chain_run(C.byref(a), None) == -1
    a.n_chains, a.n_lags, a.state, a.stats = 8, 50, 1, 1
    assert lib.mcmc_chain_run(C.byref(a), None) == -1 and b"n_lags" in lib.mcmc_last_error()
    a.n_lags = 4
    a.n_steps, a.window_start = 2, 1
    assert lib.mcmc_chain_run(C.byref(a), None) == -1
    a.n_steps = 10
    assert lib.mcmc_chain_run(C.byref(a), None) == -2  # no workspace
    x = L.XsArgs()
    assert lib.mcmc_launch_warmup_metropolis(C.byref(x), 1, None) == -1
    a.workspace_bytes = 0
    assert lib.mcmc_chain_workspace_bytes(C.byref(a)) == 1 * 1 * L.stats_len(4) * 8
    out = (C.c_double * 6)()
    assert lib.mcmc_cross_sections(None, None, None, 0, out) == -1
    # maximum size: the Philox counter carries 40 bits of chain id
    big = L.ChainArgs()
    big.n_chains, big.n_coords, big.n_steps, big.state, big.stats = (1 << 39) + 1, 1, 8, 1, 1
    assert lib.mcmc_chain_run(C.byref(big), None) == -1
    # entry points added after the first ABI: null / empty arguments are refused before any device call
    assert lib.mcmc_launch_integrate_inel(None, 1, 0.0, None, 1.0, None) == -1
    assert lib.mcmc_launch_init_sample_from_pdf(None, None) == -1
    assert lib.mcmc_run_inel_scan(None, None, None, None, 1.0) == -1
    assert lib.mcmc_init_from_pdf(None) == -1
    best = C.c_double()
    tgt = L.Target(L.TARGET_R2WS, 2.0, 0.54)
    assert lib.mcmc_tune_stepsize(C.byref(tgt), 2.0, 1.0, 5, 64, 10, 100, 1, 1.0, None, None, None, C.byref(best)) == -1
    assert lib.mcmc_tune_stepsize(C.byref(tgt), 1.0, 2.0, 5, 64, 10, 20, 1, 1.0, None, None, None, C.byref(best)) == -1
    assert b"MCMC_MAX_LAGS" in lib.mcmc_last_error()
    m = L.MvnArgs()
    assert lib.mcmc_mvn_run(C.byref(m), None) == -1


def test_product_does_not_import_the_oracle():
    """The oracle is the checker: nothing under mcmc_amd/ or include/ may name it."""
    bad = []
    for base in ("mcmc_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".txt")):
                    text = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle_lib|liboracle|mcmc_oracle|oracle/", text):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
    code = "import sys; import mcmc_amd.lib; assert not any('oracle' in m for m in sys.modules), 'oracle imported'"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


def test_missing_library_fails_loudly(tmp_path):
    """No CPU fallback: importing the binding without libmcmc_hip.so raises."""
    pkg = tmp_path / "mcmc_amd"
    pkg.mkdir()
    for f in ("__init__.py", "lib.py"):
        (pkg / f).write_text(open(os.path.join(ROOT, "mcmc_amd", f)).read())
    r = subprocess.run([sys.executable, "-c", "import mcmc_amd.lib"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr
