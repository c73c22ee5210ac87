#!/bin/bash
# tools/gpu_ab.sh LIB... -- A/B timing on the GPU box: bench.py (chain + secondary kernels) with each library build
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ab
mkdir -p "$OUT"
cd "$ROOT"
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  MCMC_HIP_LIB=$ROOT/$lib timeout -k 10 300 python -u bench.py --no-cpu-baseline > "$OUT/$tag.json" 2> "$OUT/$tag.err"
  rc=$?; [ $rc -eq 0 ] || { tail -n 20 "$OUT/$tag.err"; exit $rc; }
  python3 - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
s = d.get("secondary") or {}
print("%-24s chain %.4g (%.1f cyc)  lag49 %.4g  nucleon %.4g  stress %.4g  store %.4g  inel %.4g  sab13 %.4g" % (
    sys.argv[2], d["value"], (r.get("valu_issue") or {}).get("measured_cycles_per_wave_proposal", 0), d.get("lag49_proposals_per_sec_per_gpu", 0),
    s.get("nucleon_chain_proposals_per_s", 0), s.get("stress64_lag1000_proposals_per_s", 0), s.get("chain_writeback_proposals_per_s", 0),
    s.get("inel_pbpb_b7_event_samples_per_s", 0), s.get("sab_pbpb_b13_event_samples_per_s", 0)))
PY
done
