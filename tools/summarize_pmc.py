#!/usr/bin/env python3
"""tools/summarize_pmc.py TAG OUT.json [KERNEL_SUBSTRING [SKIP]] -- condense gpurun_out/prof_TAG (tools/gpu_profile.sh) into the JSON that
bench.py reads for roofline.traffic, and copy the kernel-trace stats next to it."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out_path = sys.argv[1], sys.argv[2]
kname = sys.argv[3] if len(sys.argv) > 3 else "chain_kernel"
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # leading dispatches of the kernel that are not bench steps (set-up launches)
base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
bench = json.loads(open(os.path.join(base, "trace.json")).read().strip().splitlines()[-1])
cfg = bench["config"]
out = {"source": f"tools/gpu_profile.sh {tag}: bench.py --steps 3 --warmup 1, one rocprofv3 run per pass "
                 "(kernel-trace; SQ+GRBM; FETCH_SIZE; WRITE_SIZE)",
       "workload": cfg["workload"], "kernel": None, "per_launch": {}}
for f in ("pmc_sq", "pmc_fetch", "pmc_write"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(base, f, f"{f}_counter_collection.csv"))):
        if kname in r["Kernel_Name"]:
            out["kernel"] = r["Kernel_Name"]
            agg[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in agg.items():
        vals = [x for _, x in sorted(v)][skip:]
        out["per_launch"][k] = sum(vals) / len(vals)
kern_ms = None
for r in csv.DictReader(open(os.path.join(base, "trace", "trace_kernel_stats.csv"))):
    if kname in r["Name"]:
        kern_ms = float(r["AverageNs"]) * 1e-6
# per launch, in launch order: the trace run uses bench.py's default 2 warm-up + 10 timed launches
launches = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
                  for r in csv.DictReader(open(os.path.join(base, "trace", "trace_kernel_trace.csv"))) if kname in r["Kernel_Name"])
launch_ms = [d for _, d in launches][skip:]
warm = bench.get("warmup", 0)
timed_ms = sum(launch_ms[warm:]) / max(len(launch_ms[warm:]), 1)
p = out["per_launch"]
props = cfg["chains_per_gpu"] * cfg["coords"] * cfg["sweeps_per_step"]
cycles = p["GRBM_GUI_ACTIVE"] / 8
out["derived"] = {
    "kernel_ms_kernel_trace": kern_ms,
    "kernel_ms_per_launch_kernel_trace": launch_ms,
    "kernel_ms_timed_launches_kernel_trace": timed_ms,
    "kernel_ms_bench_hip_events_same_run": bench["roofline"]["kernel_ms"],
    "hbm_bytes_per_launch": p["FETCH_SIZE"] * 1024 * 2 + p["WRITE_SIZE"] * 1024,
    "note_fetch": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)",
    "algorithmic_bytes_per_launch": bench["roofline"].get("algorithmic_bytes_per_launch"),
    "note_clock": "GRBM cycles come from the 4-launch PMC run, whose launches are slower (clock still ramping) than the timed ones",
    "valu_insts_per_proposal": p["SQ_INSTS_VALU"] * 64 / props,
    "simd_cycles_per_wave_proposal": cycles * 1024 / (props / 64),
    "VALUBusy_gfx9_formula_pct": 100 * p["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cycles,
}
json.dump(out, open(os.path.join(ROOT, out_path), "w"), indent=1)
shutil.copy(os.path.join(base, "trace", "trace_kernel_stats.csv"), os.path.join(ROOT, out_path.replace(".json", "_kernel_stats.csv")))
print(json.dumps(out["derived"], indent=1))
