#!/bin/bash
# tools/gpu_round2_b.sh -- GPU test suite + the xsection kernel timings (tools/bench_xs.py) after the TU split
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r2b
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1
rc=$?; tail -n 30 "$OUT/pytest_gpu.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u tools/bench_xs.py --rule nodiv > "$OUT/bench_xs.jsonl" 2> "$OUT/bench_xs.err"
rc=$?; tail -n 25 "$OUT/bench_xs.jsonl"; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench_xs.err"; exit $rc; }
timeout -k 10 300 python -u tools/bench_xs.py --rule log --which c3,c4 > "$OUT/bench_xs_log.jsonl" 2> "$OUT/bench_xs_log.err"
rc=$?; tail -n 25 "$OUT/bench_xs_log.jsonl"; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench_xs_log.err"; exit $rc; }
echo ALL_OK
