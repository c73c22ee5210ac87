#!/usr/bin/env python3
"""Generates tests/golden/macro_*.mac: OUTPUTS of the reference's macro generator (xsection/tools/MCMCMacro.py,
IntegrationSettings.GenerateCode, imported from /root/reference in the build container), for the configurations the
xsection driver's `macro=` option is tested on.  The generator script does not travel; its output files do, together
with the settings that produced them (macro_expected.json: what a reader of each file must recover).

  macro_he4_he4.mac      the reference's default collision (main.cu:38-82): 2+2 on 2+2 nucleons, one density for all
  macro_pb208_au197.mac  Pb-208 on Au-197, literature Woods-Saxon sets, protons and neutrons of a nucleus with
                         different densities (SetProtonPDFA / SetNeutronPDFA), fixed seed
  macro_c12_per_nucleon.mac  C-12 on C-12 with ONE neutron of A and ONE proton of B given their own densities
                         (SetNeutronPDFA(pdf, i) / SetProtonPDFB(pdf, i), MCMCMacro.py:172-199) and a Gaussian kind
"""
import importlib.util
import json
import os
import sys

sys.dont_write_bytecode = True   # the reference tree is read-only by agreement: no __pycache__ next to MCMCMacro.py

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/xsection/tools/MCMCMacro.py"


def load():
    spec = importlib.util.spec_from_file_location("MCMCMacro", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pdf(M, func, **params):
    return M.PDF([{"type": "float", "name": k, "value": v} for k, v in params.items()], func)


def settings(M, isoA, isoB, pdfA, pdfB, blocks, threads, samples, runs, mix, b0, db, ndbs, seed, outFile, rms=True, sab=True):
    s = M.IntegrationSettings()
    s.SetIsotopeA(isoA)
    s.SetIsotopeB(isoB)
    s.SetPDFA(pdfA)
    s.SetPDFB(pdfB)
    s.SetImpactParamterStepping(b0, db, ndbs)
    s.SetOutputDir("./")
    s.SetOutputFile(outFile)
    s.SetIntegrationTypes(rms, sab, False, False)
    s.SetMarkovMixLength(mix)
    s.SetGPUKernelParams(runs, blocks, threads, samples)
    if seed is not None:
        s.overrideSeed = True
        s.newSeed = seed
    return s


def describe(s):
    """What a reader of the macro must recover: per-nucleon (function, parameters) in pdfArr order (protons A, neutrons A,
    protons B, neutrons B: MCMCMacro.py:315-345) and the scalar settings."""
    def one(p):
        return {"func": p.funcName, "params": {q["name"]: float(q["value"]) for q in p.funcParams}}
    return {
        "nNeutronA": s.nNeutronsA, "nProtonA": s.nProtonsA, "nNeutronB": s.nNeutronsB, "nProtonB": s.nProtonsB,
        "nucleons": [one(p) for p in s.ProtonPDFListA + s.NeutronPDFListA + s.ProtonPDFListB + s.NeutronPDFListB],
        "totalBlocks": s.totalBlocks, "threadsPerBlock": s.threadsPerBlock, "samplesPerThread": s.samplesPerThread,
        "totalRuns": s.totalRuns, "samplesForMix": s.samplesForMix, "b0": s.b0, "db": s.db, "ndbs": s.ndbs,
        "seed": s.newSeed if s.overrideSeed else None, "outDir": s.outDir, "outFile": s.outFile,
        "integrateRMS": bool(s.IntegrateRMS), "integrateSAB": bool(s.IntegrateSAB),
    }


def main():
    M = load()
    expected = {}

    ws = pdf(M, "R2WoodsSaxon", c=2.0, a=0.54)
    s = settings(M, "HE-4", "HE-4", ws, ws, 256, 128, 256, 16, 8192, 0.0, 0.1, 200, None, "out.root")
    expected["macro_he4_he4.mac"] = describe(s)
    open(os.path.join(HERE, "macro_he4_he4.mac"), "w").write(s.GenerateCode())

    pb = pdf(M, "R2WoodsSaxon", c=6.62, a=0.546)
    s = settings(M, "PB-208", "197-AU", pb, pdf(M, "R2WoodsSaxon", c=6.38, a=0.535), 64, 128, 32, 2, 4096, 0.05, 0.5, 24,
                 314159265, "pbau.root")
    skin = pb.ModifyParameterValue("c", 6.69).ModifyParameterValue("a", 0.56)   # a neutron skin for Pb-208
    for i in range(s.nNeutronsA):   # (the one-argument SetNeutronPDFA is shadowed by the indexed overload, MCMCMacro.py:164-174)
        s.SetNeutronPDFA(skin, i)
    expected["macro_pb208_au197.mac"] = describe(s)
    open(os.path.join(HERE, "macro_pb208_au197.mac"), "w").write(s.GenerateCode())

    c12 = pdf(M, "R2WoodsSaxon", c=2.1, a=0.5)
    s = settings(M, "C-12", "C-12", c12, c12, 32, 128, 64, 4, 2048, 0.0, 0.25, 12, 271828, "c12.root", rms=True, sab=True)
    s.SetNeutronPDFA(pdf(M, "R2Gaussian", variance=2.5), 3)
    s.SetProtonPDFB(pdf(M, "WoodsSaxon", c=2.4, a=0.45), 0)
    expected["macro_c12_per_nucleon.mac"] = describe(s)
    open(os.path.join(HERE, "macro_c12_per_nucleon.mac"), "w").write(s.GenerateCode())

    json.dump(expected, open(os.path.join(HERE, "macro_expected.json"), "w"), indent=1)
    print("wrote", ", ".join(sorted(expected)), "and macro_expected.json")


if __name__ == "__main__":
    main()
