#!/bin/bash
# tools/gpu_profile_store.sh -- HBM traffic of the chain kernel WITH sample write-back (tools/bench_store.py) from the
# PMC counters: WRITE_SIZE and FETCH_SIZE, each in a run of its own.  Output: gpurun_out/prof_store/*.csv
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_store
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
    echo "== $c"
    timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -o $c -- python3 $ROOT/tools/bench_store.py --reps 3 > "$OUT/$c.json" 2> "$OUT/$c.err" || exit 1
    tail -n 1 "$OUT/$c.json" | cut -c1-160
done
