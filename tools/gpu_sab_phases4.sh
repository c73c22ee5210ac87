#!/bin/bash
# tools/gpu_sab_phases4.sh -- on the GPU box: the inelastic-only S_AB mode built with -DMCMC_SAB_PROFILE=1..4
# (tools/bin/libprof4_N.so): `p_inel` of tools/bench_xs.py then reads wall cycles per event sample of one phase
# (1 MH phase, 2 lazy angles + probes, 3 pair sweeps, 4 everything).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sabprof4
mkdir -p "$OUT"
cd "$ROOT"
for v in 1 2 3 4; do
MCMC_HIP_LIB=$ROOT/tools/bin/libprof4_$v.so timeout -k 10 200 python -u tools/bench_xs.py --which c4 --rule nodiv --bs 0 --inel-bs 0,7,14,18 > "$OUT/v$v.jsonl" 2> "$OUT/v$v.err"
rc=$?; [ $rc -eq 0 ] || { tail -n 20 "$OUT/v$v.err"; exit $rc; }
python3 - "$OUT/v$v.jsonl" $v <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if d["kernel"] == "MCIntegrate_Inel":
        print("phase %s  b %4.1f  ms %.4f  wall cycles per event-sample %.0f" % (sys.argv[2], d["b"], d["ms"], d["p_inel"]))
PY
done
