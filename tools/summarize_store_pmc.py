#!/usr/bin/env python3
"""tools/summarize_store_pmc.py [DIR ...] -> profiles/r2_store_path_pmc.json: the counters of tools/gpu_store_pmc.sh per kernel
(mean over the dispatches of equal size: the stress bench's ring-priming launch is dropped by taking the most frequent
instruction count), with the ratios the write-back analysis of DESIGN 4.1 / 4.4 quotes."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIRS = sys.argv[1:] or ["store_pmc_before", "store_pmc"]
out = {"source": "tools/gpu_store_pmc.sh (rocprofv3 --pmc, one run per counter set) on tools/bench_mvn.py and tools/bench_store.py",
       "units": "sums over the chip; TCP_* over 256 TCPs, SQ_* over the SQs, GRBM_GUI_ACTIVE over 8 XCDs"}
for d in DIRS:
    base = os.path.join(ROOT, "gpurun_out", d)
    res = {}
    for tag in ("mvn", "store"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for ps in ("tlb", "sq", "tcp", "tcc"):
            f = os.path.join(base, f"{tag}_{ps}", f"{tag}_{ps}_counter_collection.csv")
            if not os.path.exists(f):
                continue
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "mvn_reg_kernel" in k or "mvn_kernel" in k or "chain_kernel" in k:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            ent = {}
            for c, v in cs.items():
                v = sorted(v)
                ent[c] = v[len(v) // 2]          # median: the priming launch (longer) is one of >= 5 dispatches
            if ent.get("SQ_INSTS_VMEM_WR"):
                ent["ratios"] = {
                    "utcl1_miss_per_request": ent.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / max(ent.get("TCP_UTCL1_REQUEST_sum", 1), 1),
                    "ta_cmd_fifo_full_over_sq_busy": ent.get("SQ_VMEM_TA_CMD_FIFO_FULL", 0) / max(ent.get("SQ_BUSY_CYCLES", 1), 1),
                    "tcp_pending_stall_per_tcp_over_cycles": ent.get("TCP_PENDING_STALL_CYCLES_sum", 0) / 256 / max(ent.get("GRBM_GUI_ACTIVE", 1) / 8, 1),
                    "write_requests_64B_per_store_inst": ent.get("TCP_TCC_WRITE_REQ_sum", 0) / ent["SQ_INSTS_VMEM_WR"],
                    "wait_inst_any_over_wave_cycles": ent.get("SQ_WAIT_INST_ANY", 0) / max(ent.get("SQ_WAVE_CYCLES", 1), 1)}
            res[k] = ent
    out[d] = res
path = os.path.join(ROOT, "profiles", "r2_store_path_pmc.json")
json.dump(out, open(path, "w"), indent=1)
for d in DIRS:
    for k, e in out[d].items():
        if "ratios" in e:
            print(d, k[:60], {a: round(b, 3) for a, b in e["ratios"].items()})
