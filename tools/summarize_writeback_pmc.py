#!/usr/bin/env python3
"""tools/summarize_writeback_pmc.py -> profiles/r3_writeback_limiter.json: the counters of tools/gpu_writeback_pmc.sh per case
(median over the dispatches of the chain kernel), with the shader clock each case sustained (GRBM_GUI_ACTIVE / 8 XCDs over
the dispatch's duration) and the cycles a wave-store adds."""
import collections
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(ROOT, "gpurun_out", "wb_pmc")
out = {"source": "tools/gpu_writeback_pmc.sh (rocprofv3 --pmc, one run per counter set) on tools/bench_writeback.py 256 <case>: "
                 "2^24 chains x 3 coordinates x 256 sweeps, CROSS rule, FAST math",
       "units": "sums over the chip per launch; *_sum over all TCPs / TCC channels; GRBM_GUI_ACTIVE over 8 XCDs",
       "cases": {}}
props = (1 << 24) * 3 * 256
for case in ("none", "rows", "pairs"):
    ent = {}
    for ps in ("sq", "tcp", "tcc"):
        f = os.path.join(base, f"{case}_{ps}", f"{case}_{ps}_counter_collection.csv")
        if not os.path.exists(f):
            continue
        agg = collections.defaultdict(list)
        dur = {}
        for r in csv.DictReader(open(f)):
            if "chain_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
                ent["kernel"] = r["Kernel_Name"]
        for c, v in agg.items():
            v = sorted(v)
            ent[c] = v[len(v) // 2]
        if ps == "sq" and dur:
            d = sorted(dur.values())
            ent["kernel_ms_under_pmc"] = d[len(d) // 2]
    if "GRBM_GUI_ACTIVE" in ent:
        cyc = ent["GRBM_GUI_ACTIVE"] / 8
        ent["derived"] = {
            "shader_clock_GHz": cyc / (ent["kernel_ms_under_pmc"] * 1e6),
            "simd_cycles_per_wave_proposal": cyc * 1024 / (props / 64),
            "valu_insts_per_proposal": ent["SQ_INSTS_VALU"] * 64 / props,
            "wait_inst_any_over_wave_cycles": ent["SQ_WAIT_INST_ANY"] / ent["SQ_WAVE_CYCLES"],
        }
        if ent.get("SQ_INSTS_VMEM_WR"):
            ent["derived"]["store_insts_per_proposal"] = ent["SQ_INSTS_VMEM_WR"] * 64 / props
            ent["derived"]["mean_cycles_a_store_is_in_flight"] = ent["SQ_INST_LEVEL_VMEM"] / ent["SQ_INSTS_VMEM_WR"]
        if ent.get("TCP_TCC_WRITE_REQ_sum"):
            ent["derived"]["tcp_to_tcc_write_latency_cycles"] = ent["TCP_TCC_WRITE_REQ_LATENCY_sum"] / ent["TCP_TCC_WRITE_REQ_sum"]
            ent["derived"]["tcp_pending_stall_per_tcp_over_cycles"] = ent["TCP_PENDING_STALL_CYCLES_sum"] / 256 / cyc
        if ent.get("TCC_EA0_WRREQ_sum"):
            ent["derived"]["ea_write_latency_cycles"] = ent["TCC_EA0_WRREQ_LEVEL_sum"] / ent["TCC_EA0_WRREQ_sum"]
            ent["derived"]["ea_wrreq_64B_fraction"] = ent["TCC_EA0_WRREQ_64B_sum"] / ent["TCC_EA0_WRREQ_sum"]
            ent["derived"]["ea_wrreq_stall_per_channel_over_cycles"] = ent["TCC_EA0_WRREQ_STALL_sum"] / 128 / cyc
    out["cases"][case] = ent
c = out["cases"]
if "derived" in c.get("none", {}) and "derived" in c.get("rows", {}):
    for k in ("rows", "pairs"):
        if "derived" in c.get(k, {}):
            d = c[k]["derived"]
            d["cycles_added_per_wave_proposal_vs_none"] = d["simd_cycles_per_wave_proposal"] - c["none"]["derived"]["simd_cycles_per_wave_proposal"]
            d["time_ratio_vs_none"] = c[k]["kernel_ms_under_pmc"] / c["none"]["kernel_ms_under_pmc"]
            d["clock_ratio_vs_none"] = d["shader_clock_GHz"] / c["none"]["derived"]["shader_clock_GHz"]
json.dump(out, open(os.path.join(ROOT, "profiles", "r3_writeback_limiter.json"), "w"), indent=1)
for k, e in c.items():
    print(k, json.dumps(e.get("derived"), indent=0))
