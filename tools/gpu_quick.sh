#!/bin/bash
# tools/gpu_quick.sh [bench args] -- on the GPU box: the GPU test suite, then the bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/quick
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1
rc=$?; tail -n 15 "$OUT/pytest_gpu.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
rc=$?; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench.err"; exit $rc; }
python3 - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g  kernel_ms %.3f  frac %.4f  valu_issue %s" % (d["value"], r["kernel_ms"], r["frac"], json.dumps(r.get("valu_issue"))))
print("lag49 %s  tau %s" % (d.get("lag49_proposals_per_sec_per_gpu"), d.get("tau_int")))
print(json.dumps(d.get("secondary")))
PY
