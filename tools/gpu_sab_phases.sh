#!/bin/bash
# tools/gpu_sab_phases.sh -- on the GPU box: the S_AB kernel (product + counters, Pb-Pb) built with -DMCMC_SAB_PROFILE
# (tools/bin/libmcmc_hip_sabprof.so): per-event shader-clock totals of its phases come back in place of the counters.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sabprof
mkdir -p "$OUT"
cd "$ROOT"
for b in 0 4 7 10 13; do
MCMC_HIP_LIB=$ROOT/tools/bin/libmcmc_hip_sabprof.so timeout -k 10 200 python -u tools/bench_xs.py --which c4 --rule nodiv --bs $b --inel-bs 0 --hits-b $b > "$OUT/b$b.jsonl" 2> "$OUT/b$b.err"
rc=$?; [ $rc -eq 0 ] || { tail -n 20 "$OUT/b$b.err"; exit $rc; }
python3 - "$OUT/b$b.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if d["kernel"] == "MCIntegrate_S_AB+hits":
        mh, cull, pair, n = d["hits"]
        ev = d["events"]
        tot = d["S_AB"] * ev
        print("b %.0f cull %s ms %.3f  wall cycles per event-sample: total %.0f  mh %.0f  cull+marks %.0f  pairs %.0f  other %.0f" % (
            d["b"], d["cull"], d["ms"], tot / n, mh / n, cull / n, pair / n, (tot - mh - cull - pair) / n))
PY
done
