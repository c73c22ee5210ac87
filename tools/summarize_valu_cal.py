#!/usr/bin/env python3
"""tools/summarize_valu_cal.py [OUT.json] -- condense gpurun_out/valu_cal (tools/gpu_valu_calibration.sh) into
profiles/r2_valu_busy_calibrated.json: what the SQ counters read on pure-instruction loops whose VALU issue occupancy is
100 % by construction, the issue cost per instruction those loops give (SIMD cycles per wave64 instruction, clock from
GRBM_GUI_ACTIVE of the same dispatch), and -- with that calibration -- the VALU issue occupancy of the kernels of the path.

Finding the calibration rests on: SQ_ACTIVE_INST_VALU tallies ONE quad-cycle per VALU instruction (two per
transcendental) whatever the instruction's issue cost, so the gfx9 `VALUBusy` formula reads 92-175 % at full issue and
is no occupancy measure.  Occupancy here = sum over instruction classes of (dynamic count x calibrated cost) / SIMD cycles,
with the dynamic counts from SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 / _INT32 / _INT64 / _CVT; instructions those class
counters do not tally (bitwise ops, moves, compares, selects, min/max: `other`) are priced between the cheapest (2.29) and
the dearest (4.33) non-transcendental cost, and -- for the chain kernels -- by the static mix of tools/isa_mix.py."""
import collections
import csv
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = os.path.join(ROOT, "gpurun_out", "valu_cal")
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r2_valu_busy_calibrated.json")
SIMDS = 1024
CLASSES = ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT")


def load(tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    path = os.path.join(BASE, tag, f"{tag}_counter_collection.csv")
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def trace(tag):
    path = os.path.join(BASE, tag, f"{tag}_kernel_stats.csv")
    return {r["Name"]: float(r["AverageNs"]) * 1e-6 for r in csv.DictReader(open(path))} if os.path.exists(path) else {}


def main():
    out = {"source": "tools/gpu_valu_calibration.sh on one MI355X: rocprofv3 --pmc, one run per counter set (A: SQ_WAVES "
                     "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY "
                     "SQ_INSTS_SALU GRBM_GUI_ACTIVE; B: SQ_INSTS_VALU + the seven SQ_INSTS_VALU_* class counters; "
                     "C: LDS counters), plus a --kernel-trace --stats run; programs after `--`: tools/bin/ubench_cal, "
                     "bench.py, tools/bench_mvn.py, tools/bench_xs.py",
           "units": "cycles = GRBM_GUI_ACTIVE / 8 (sum over the 8 XCDs); SQ_* summed over the chip; 1024 SIMDs",
           "calibration": {}, "kernels": {}}
    a, b = load("ubench_a"), load("ubench_b")
    cost = {}
    for k in a:
        op = re.sub(r"void cal<op_(\w+)>.*", r"\1", k)
        A, B = a[k], b.get(k, {})
        cyc = A["GRBM_GUI_ACTIVE"] / 8
        cls = [c for c in CLASSES if B.get("SQ_INSTS_VALU_" + c, 0) > 0.5 * B.get("SQ_INSTS_VALU", 1)]
        n_per = {"cmp_cnd3": 4, "cmp_fma_cnd3": 12, "fma12": 12, "cmpx_mov3": 4, "sub_ashr_bitop3": 5, "cmp_sgpr_cnd3": 4}.get(op, 1)
        cost[op] = cyc * SIMDS / A["SQ_INSTS_VALU"]
        out["calibration"][op] = {
            "cycles_per_wave_inst": cost[op], "insts_in_op": n_per,
            "SQ_ACTIVE_INST_VALU_per_inst": A["SQ_ACTIVE_INST_VALU"] / A["SQ_INSTS_VALU"],
            "VALUBusy_gfx9_formula_pct_at_full_issue": 100 * A["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * SIMDS),
            "SQ_WAIT_INST_ANY_over_SQ_WAVE_CYCLES": A["SQ_WAIT_INST_ANY"] / A["SQ_WAVE_CYCLES"],
            "class_counter": cls[0] if cls else "none (other)"}
    c_cls = {"ADD_F32": cost.get("add", 2.31), "MUL_F32": cost.get("mul", 2.30), "FMA_F32": 0.5 * (cost.get("fma", 2.73) + cost.get("fmac", 2.67)),
             "TRANS_F32": cost.get("exp", 8.13), "INT64": cost.get("mad64", 4.33), "INT32": 2.5, "CVT": 4.2}
    other_lo, other_hi = cost.get("mov", 2.29), cost.get("mad64", 4.33)
    out["class_costs_used"] = dict(c_cls, other_low=other_lo, other_high=other_hi,
                                   note="INT32 / CVT not calibrated by a loop of their own: 2.5 / 4.2 assumed (both < 6 % of any kernel)")

    for tag, want in (("chain", "chain_kernel"), ("lag49", "chain_kernel"), ("stress", "mvn_(reg_)?kernel"), ("sab", "sab_kernel|xs_chain_kernel")):
        A_, B_, C_, T_ = load(tag + "_a"), load(tag + "_b"), load(tag + "_c"), trace(tag + "_trace")
        for k, A in A_.items():
            if not re.search(want, k) or A.get("SQ_INSTS_VALU", 0) < 1e6:
                continue
            B, C = B_.get(k, {}), C_.get(k, {})
            cyc = A["GRBM_GUI_ACTIVE"] / 8
            n = A["SQ_INSTS_VALU"]
            frac = {c: B.get("SQ_INSTS_VALU_" + c, 0.0) / max(B.get("SQ_INSTS_VALU", 1.0), 1.0) for c in CLASSES}
            other = max(0.0, 1.0 - sum(frac.values()))
            known = sum(frac[c] * c_cls[c] for c in CLASSES)
            ent = {"kernel_ms_kernel_trace": T_.get(k), "clock_GHz_GRBM": cyc / (T_[k] * 1e-3) / 1e9 if T_.get(k) else None,
                   "SQ_INSTS_VALU": n, "simd_cycles_per_valu_inst": cyc * SIMDS / n,
                   "SQ_ACTIVE_INST_VALU_per_inst": A["SQ_ACTIVE_INST_VALU"] / n,
                   "VALUBusy_gfx9_formula_pct": 100 * A["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * SIMDS),
                   "class_fractions": dict(frac, other=other),
                   "SQ_INSTS_SALU_per_valu_inst": A.get("SQ_INSTS_SALU", 0) / n,
                   "SQ_WAIT_INST_ANY_over_SQ_WAVE_CYCLES": A["SQ_WAIT_INST_ANY"] / A["SQ_WAVE_CYCLES"],
                   "waves": A.get("SQ_WAVES"),
                   "calibrated_issue_cycles_per_inst_low": known + other * other_lo,
                   "calibrated_issue_cycles_per_inst_high": known + other * other_hi}
            ent["valu_issue_occupancy_low"] = ent["calibrated_issue_cycles_per_inst_low"] / ent["simd_cycles_per_valu_inst"]
            ent["valu_issue_occupancy_high"] = ent["calibrated_issue_cycles_per_inst_high"] / ent["simd_cycles_per_valu_inst"]
            if C:
                ent["lds"] = {"SQ_INSTS_LDS_per_valu_inst": C.get("SQ_INSTS_LDS", 0) / max(C.get("SQ_INSTS_VALU", 1), 1),
                              "SQ_LDS_BANK_CONFLICT_over_SQ_LDS_IDX_ACTIVE": C.get("SQ_LDS_BANK_CONFLICT", 0) / max(C.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                              "SQ_WAIT_INST_LDS_over_SQ_WAVE_CYCLES": C.get("SQ_WAIT_INST_LDS", 0) / max(C.get("SQ_WAVE_CYCLES", 1), 1),
                              "SQ_ACTIVE_INST_LDS_over_SQ_WAVE_CYCLES": C.get("SQ_ACTIVE_INST_LDS", 0) / max(C.get("SQ_WAVE_CYCLES", 1), 1)}
            if tag == "stress":
                ent["note"] = ("averages run over every dispatch of tools/bench_mvn.py, the ring-priming launch of 1024 sweeps included: "
                               "the per-instruction ratios hold, kernel_ms_kernel_trace is NOT the time of a 256-sweep launch (profiles/r2_mvn_stress_v3.jsonl has that)")
            out["kernels"][f"{tag}: {k}"] = ent

    # chain kernels: price `other` by the static hot-loop mix (tools/isa_mix.py) instead of the low/high bracket
    inst_cost = {"v_bitop3_b32": cost.get("bitop3"), "v_mad_u64_u32": cost.get("mad64"), "v_add_f32_e32": cost.get("add"),
                 "v_sub_f32_e32": cost.get("add"), "v_mul_f32_e32": cost.get("mul"), "v_mul_f32_e64": cost.get("mulabs"),
                 "v_mov_b32_e32": cost.get("mov"), "v_alignbit_b32": cost.get("alignbit"), "v_perm_b32": cost.get("perm"),
                 "v_fma_f32": cost.get("fma"), "v_fmac_f32_e32": cost.get("fmac"), "v_cmpx_ge_f32_e32": cost.get("cmp"),
                 "v_cmp_ge_f32_e32": cost.get("cmp"), "v_xor_b32_e32": cost.get("xor")}
    for nlag, tag in ((0, "chain"), (49, "lag49")):
        try:
            mix = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), str(nlag)], check=True,
                                            capture_output=True, text=True).stdout)
        except Exception as e:  # noqa: BLE001
            out.setdefault("static_mix_errors", []).append(f"{tag}: {e}")
            continue
        tot = 0.0
        for op, cnt in mix["histogram"].items():
            if not op.startswith("v_"):
                continue
            c = inst_cost.get(op) or (cost.get("exp") if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", op) else
                                      (cost.get("mad64") if re.match(r"v_(cmp|max|min|mad_u64|mul_hi|mul_lo)", op) else cost.get("mov")))
            tot += cnt * c
        per_prop = tot / mix["proposals_per_iteration"]
        for k, ent in out["kernels"].items():
            if k.startswith(tag + ":") and "chain_kernel" in k:
                props = {"chain": (1 << 24) * 3 * 256, "lag49": (1 << 22) * 3 * 256}[tag]
                measured = ent["simd_cycles_per_valu_inst"] * ent["SQ_INSTS_VALU"] / (props / 64)
                ent["static_mix"] = {"valu_insts_per_proposal_static_hot_loop": mix["valu_insts_per_proposal"],
                                     "valu_insts_per_proposal_dynamic": ent["SQ_INSTS_VALU"] * 64 / props,
                                     "calibrated_issue_cycles_per_wave_proposal": per_prop,
                                     "measured_simd_cycles_per_wave_proposal": measured,
                                     "valu_issue_occupancy": per_prop / measured}
        json.dump(mix, open(os.path.join(ROOT, "profiles", f"r2_chain_kernel_isa_mix{'_lag49' if nlag else ''}.json"), "w"), indent=1)
    json.dump(out, open(OUT, "w"), indent=1)
    for tag in ("chain", "lag49", "stress", "sab", "ubench"):
        src = os.path.join(BASE, tag + "_trace", f"{tag}_trace_kernel_stats.csv")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(ROOT, "profiles", f"r2_kernel_stats_{tag}.csv"))
    for k, e in out["kernels"].items():
        sm = e.get("static_mix", {})
        print("%-90s occ %.2f..%.2f %s" % (k[:90], e["valu_issue_occupancy_low"], e["valu_issue_occupancy_high"],
                                          ("static %.3f (%.1f / %.1f cyc)" % (sm["valu_issue_occupancy"], sm["calibrated_issue_cycles_per_wave_proposal"],
                                                                              sm["measured_simd_cycles_per_wave_proposal"])) if sm else ""))


if __name__ == "__main__":
    main()
