#!/bin/bash
# tools/gpu_lag_mfma_pmc.sh -- on the GPU box: matrix-pipe counters of the 49-lag chain kernel (chain_lag_kernel, v_mfma_f32_16x16x32_f16)
# over 1024-sweep launches of 2^22 chains x 3 coordinates (tools/lag49_probe.py), one rocprofv3 --pmc run.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lagpmc
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/p" -o p -- python3 "$ROOT/tools/lag49_probe.py" --sweeps 1024 --batch 2 --reps 2 --no-clock > "$OUT/p.txt" 2>&1 || { tail -5 "$OUT/p.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
f = glob.glob(f"{out}/p/**/*counter_collection.csv", recursive=True)[0]
disp = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "chain_lag_kernel" in r["Kernel_Name"]:
        disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
top = max(d["SQ_INSTS_MFMA"] for d in disp.values())
full = [d for d in disp.values() if d["SQ_INSTS_MFMA"] > 0.9 * top]      # the 1024-sweep launches
per = {k: sum(d[k] for d in full) / len(full) for k in full[0]}
chains, coords, sweeps = 1 << 22, 3, 1024
waves = chains * coords / 64
grbm_per_xcd = per["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
d = {"kernel": "mcmc::chain_lag_kernel (2^22 chains x 3 coordinates x 1024 sweeps per launch)", "launches_averaged": len(full), "per_launch": per,
     "mfma_insts_per_wave_step": per["SQ_INSTS_MFMA"] / (waves * sweeps),
     "valu_insts_per_wave_step_incl_mfma": per["SQ_INSTS_VALU"] / (waves * sweeps),
     "mfma_f16_flop_per_launch": per["SQ_INSTS_VALU_MFMA_MOPS_F16"] * 512,
     "shader_cycles_per_launch": grbm_per_xcd,
     "MfmaUtil_pct": 100.0 * per["SQ_VALU_MFMA_BUSY_CYCLES"] / (grbm_per_xcd * 1024),   # rocprofv3's MfmaUtil: busy cycles / (cycles x 1024 SIMDs)
     "mfma_busy_cycles_per_mfma": per["SQ_VALU_MFMA_BUSY_CYCLES"] / per["SQ_INSTS_MFMA"]}
print(json.dumps(d, indent=1))
PY
