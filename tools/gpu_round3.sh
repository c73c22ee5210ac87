#!/bin/bash
# tools/gpu_round3.sh bench|prof|prof2 -- on the GPU box: the round-3 record.  bench: GPU test suite, the bench workloads (default,
# nucleon, stress, stress --full), the two-rank gloo rehearsal, per-kernel timings.  prof: kernel trace + PMC passes of bench.py
# (default workload), the 49-lag kernel's kernel trace.  prof2: the same passes for `--workload stress --full`.
# Every GPU step under its own time limit; the script stops at the first failure.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
step() { # name, limit, command... (stdout -> $OUT/name.out, stderr -> $OUT/name.err)
    local name=$1 limit=$2; shift 2
    echo "== $name"
    timeout -k 10 "$limit" "$@" > "$OUT/$name.out" 2> "$OUT/$name.err"
    local rc=$?
    tail -c 600 "$OUT/$name.out"; echo
    [ $rc -eq 0 ] || { echo "rc=$rc"; tail -n 20 "$OUT/$name.err"; exit $rc; }
}
PART=${1:-bench}
if [ "$PART" = bench ]; then
step pytest_gpu 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread
step bench_n1 400 python -u bench.py
step bench_nucleon 300 python -u bench.py --workload nucleon
step bench_stress 300 python -u bench.py --workload stress
step bench_stress_full 300 python -u bench.py --workload stress --full
MCMC_BENCH_BACKEND=gloo step bench_n2_gloo 300 python -u bench.py --gpus 2 --chains 4194304 --steps 5 --warmup 1
step bench_xs 300 python -u tools/bench_xs.py --rule nodiv
step bench_mvn 300 python -u tools/bench_mvn.py
elif [ "$PART" = prof ]; then
step profile 900 bash tools/gpu_profile.sh r3
export TMPDIR=/tmp
mkdir -p "$OUT/lag49"
( cd /tmp && step lag49_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/lag49" -o lag49 -- python3 "$ROOT/tools/lag49_probe.py" --sweeps 1024 --batch 4 --reps 3 --no-clock ) || exit 1
find "$OUT/lag49" -name "*.csv"
else
step profile_stress_full 900 bash tools/gpu_profile.sh r3sf --workload stress --full
fi
echo ALL_OK
