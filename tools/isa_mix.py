#!/usr/bin/env python3
"""tools/isa_mix.py -- static instruction mix of the benchmarked chain kernel's hot loop (one group of 8
proposals), from hipcc -S of mcmc_amd/csrc/chain_kind.hip with the Makefile's flags.  Gives the VALU issue
cycles per wave-proposal that bench.py sets against the measured rate.  Issue costs per wave64 instruction
(SIMD cycles), from tools/ubench2.hip on MI355X (profiles/r1_valu_issue_costs.txt, anchored on v_exp_f32 = 8):
    ideal model   : 2 for full-rate ops (v_bitop3_b32 included), 4 for the class that measures 4.0-4.3 with
                    VGPR operands only (v_mad_u64_u32, v_alignbit_b32, v_perm_b32, v_max_f32, v_cmp_*, ...),
                    8 for transcendentals
    isolated costs: what each instruction showed alone (full-rate ops 2.3-2.9 instead of 2)
The kernel cannot beat the ideal model; the second is what the same instructions achieve in a loop of their own.
Runs in the build container (no GPU needed)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NLAG = int(sys.argv[1]) if len(sys.argv) > 1 else 0
# float, R2WS, CROSS, FAST, NLAG lags, Philox draws, no write-back, no stamps
KERNEL = f"_ZN4mcmc12chain_kernelIfLi1ELi4ELi1ELi{NLAG}ELb0ELi0ELb0EEEvNS_11ChainParamsE"
QUARTER = re.compile(r"^v_(exp_f32|log_f32|rcp_f32|rsq_f32|sqrt_f32|sin_f32|cos_f32)")
HALF = re.compile(r"^v_(mad_u64_u32|mul_hi_u32|mul_lo_u32|alignbit_b32|perm_b32|bfi_b32|xad_u32|add3_u32|max_f32|min_f32|cmp_|cmpx_)")
# v_cndmask_b32_e32 (mask in VCC): 7.7 cycles in the v_cmp + 3 v_cndmask pattern (tools/ubench_cal.hip, profiles/r2_ubench_cal.txt)
SLOW = {"v_cndmask_b32_e32": 7.7}
ISOLATED = {"v_xor_b32_e32": 2.93, "v_bitop3_b32": 2.24, "v_add_f32_e32": 2.57, "v_sub_f32_e32": 2.43, "v_mul_f32_e32": 2.31, "v_fma_f32": 2.89,
            "v_fmac_f32_e32": 2.80, "v_alignbit_b32": 4.14, "v_perm_b32": 4.14, "v_mad_u64_u32": 4.29, "v_max_f32_e32": 4.08,
            "v_cmp_ge_f32_e32": 4.29}  # cycles; anything else: its class in the ideal model


def ideal_cost(op):
    if op in SLOW:
        return SLOW[op]
    return 8.0 if QUARTER.match(op) else 4.0 if HALF.match(op) else 2.0


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-I" + os.path.join(ROOT, "include"), "-DMCMC_BUILD", "-fno-slp-vectorize", "-DMCMC_KIND=1", "-S",
                        "--cuda-device-only", "-o", out, os.path.join(ROOT, "mcmc_amd/csrc/chain_kind.hip")],
                       check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    a = lines.index(next(l for l in lines if l.startswith(KERNEL + ":")))
    b = next(i for i in range(a, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[a:b]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = body[labels[m.group(1)]:i + 1]
            if any(re.match(r"^\.LBB", x) for x in seg[1:]):
                continue  # not innermost
            n_mad = sum("v_mad_u64_u32" in x for x in seg)
            if best is None or n_mad > best[0]:
                best = (n_mad, seg)
    hist = collections.Counter()
    for x in best[1]:      # inline-asm blocks arrive as one line with embedded newlines: split them
        for y in x.split("\n"):
            y = y.strip()
            if y and not y.startswith(";") and not y.startswith(".") and not y.endswith(":"):
                hist[y.split()[0]] += 1
    valu = {k: v for k, v in hist.items() if k.startswith("v_")}
    n_valu = sum(valu.values())
    n_q = sum(v for k, v in valu.items() if QUARTER.match(k))
    n_h = sum(v for k, v in valu.items() if HALF.match(k))
    ideal = sum(v * ideal_cost(k) for k, v in valu.items())
    isolated = sum(v * ISOLATED.get(k, ideal_cost(k)) for k, v in valu.items())
    proposals = 8
    res = {"kernel": KERNEL, "loop": "one group of 8 proposals (3 Philox4x32-10 blocks)", "proposals_per_iteration": proposals,
           "valu_insts": n_valu, "quarter_rate_insts": n_q, "half_rate_class_insts": n_h, "salu_insts": sum(v for k, v in hist.items() if k.startswith("s_")),
           "valu_insts_per_proposal": n_valu / proposals, "quarter_rate_per_proposal": n_q / proposals,
           "issue_cycles_per_wave_proposal_ideal": ideal / proposals,
           "issue_cycles_per_wave_proposal_isolated_costs": isolated / proposals,
           "philox_insts": valu.get("v_mad_u64_u32", 0) + valu.get("v_xor_b32_e32", 0) + valu.get("v_bitop3_b32", 0),
           "histogram": dict(sorted(hist.items(), key=lambda kv: -kv[1]))}
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
