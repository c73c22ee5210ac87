#!/bin/bash
# tools/gpu_xs.sh [bench_xs args] -- on the GPU box: the xsection GPU tests, then the C4 kernel timings
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/xs
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -u -m pytest tests/test_gpu_xsection.py -m gpu -x -q --timeout 300 --timeout-method thread > "$OUT/pytest.log" 2>&1
rc=$?; tail -n 15 "$OUT/pytest.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u tools/bench_xs.py --which c4 --rule nodiv "$@" > "$OUT/bench_xs.jsonl" 2> "$OUT/bench_xs.err"
rc=$?; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench_xs.err"; exit $rc; }
python3 - "$OUT/bench_xs.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k in ("kernel", "b", "cull", "ms", "event_samples_per_s", "hits")})
PY
