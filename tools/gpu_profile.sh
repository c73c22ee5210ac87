#!/bin/bash
# tools/gpu_profile.sh TAG [bench args...] -- on the GPU box: kernel trace + PMC passes of bench.py.
# Writes gpurun_out/prof_TAG/{trace,pmc_sq,pmc_fetch,pmc_write}/ (CSV) and gpurun_out/prof_TAG/bench.json.
# Each rocprofv3 pass is its own run (no counter multiplexing) under its own time limit.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
STEPS="--steps 3 --warmup 1"
bench_cmd() { echo "python3 $ROOT/bench.py $STEPS --no-cpu-baseline --no-ess --no-secondary --no-clock $EXTRA"; }
EXTRA="$*"

run() { # name, rocprof args...
    local name=$1; shift
    echo "== $name"
    timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- $(bench_cmd) > "$OUT/$name.json" 2> "$OUT/$name.err"
    local rc=$?
    echo "rc=$rc"; tail -n 2 "$OUT/$name.json"
    return $rc
}

# the kernel trace runs bench.py's DEFAULT step counts (2 warm-up + 10 timed launches), so that its per-launch durations are
# those of the command the bench line comes from (a 4-launch run is still ramping the clock: 12.2, 11.3, 10.8, 10.75 ms)
STEPS="--steps 10 --warmup 2"
run trace --kernel-trace --stats || exit 1
STEPS="--steps 3 --warmup 1"
run pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE || exit 1
find "$OUT" -name "*.csv" | head -40
