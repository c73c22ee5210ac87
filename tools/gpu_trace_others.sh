#!/bin/bash
# tools/gpu_trace_others.sh -- rocprofv3 kernel-trace statistics of the xsection and stress kernels
# (tools/bench_xs.py, tools/bench_mvn.py).  Output: gpurun_out/prof_others/{xs,mvn}/*_kernel_stats.csv
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_others
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/xs" -o xs -- python3 $ROOT/tools/bench_xs.py --which c3,c4 --rule cross > "$OUT/xs.jsonl" 2> "$OUT/xs.err" || exit 1
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mvn" -o mvn -- python3 $ROOT/tools/bench_mvn.py > "$OUT/mvn.jsonl" 2> "$OUT/mvn.err" || exit 1
find "$OUT" -name "*kernel_stats.csv"
