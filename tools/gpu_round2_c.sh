#!/bin/bash
# tools/gpu_round2_c.sh bench|prof -- on the GPU box: the round-2 record, in two calls (each within gpurun's limit).  GPU test suite, the three bench workloads, the two-rank
# gloo rehearsals, per-kernel timings (xsection, stress), then the profile passes (kernel trace + PMC of bench.py, the
# counter calibration).  Every GPU step under its own time limit; the script stops at the first failure.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r2c
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
step bench_n1 300 python -u bench.py
step bench_nucleon 300 python -u bench.py --workload nucleon
step bench_stress 300 python -u bench.py --workload stress
MCMC_BENCH_BACKEND=gloo step bench_n2_gloo 300 python -u bench.py --gpus 2 --chains 4194304 --steps 5 --warmup 1
MCMC_BENCH_BACKEND=gloo step bench_stress_n2_gloo 300 python -u bench.py --gpus 2 --workload stress --chains 524288 --steps 3 --warmup 1
step bench_xs 300 python -u tools/bench_xs.py --rule nodiv
step bench_xs_log 300 python -u tools/bench_xs.py --rule log --which c3,c4
step bench_mvn 300 python -u tools/bench_mvn.py
step bench_mvn16 300 python -u tools/bench_mvn.py --dim 16
else
step profile 900 bash tools/gpu_profile.sh r2v3
step valu_cal 1100 bash tools/gpu_valu_calibration.sh
fi
echo ALL_OK
