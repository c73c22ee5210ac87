#!/bin/bash
# tools/gpu_round2_a.sh -- on the GPU box: the GPU test suite, the no-alloc test, the bench line (N = 1), a two-rank
# gloo rehearsal of `bench.py --gpus 2` (ranks started by bench.py itself) and the nucleon / stress workloads.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r2a
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1
rc=$?; tail -n 5 "$OUT/pytest_gpu.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_noalloc.log" 2>&1
rc=$?; tail -n 5 "$OUT/pytest_noalloc.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
rc=$?; tail -c 1500 "$OUT/bench_n1.json"; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench_n1.err"; exit $rc; }
MCMC_BENCH_BACKEND=gloo timeout -k 10 300 python -u bench.py --gpus 2 --chains 4194304 --steps 5 --warmup 1 > "$OUT/bench_n2_gloo.json" 2> "$OUT/bench_n2_gloo.err"
rc=$?; tail -c 800 "$OUT/bench_n2_gloo.json"; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench_n2_gloo.err"; exit $rc; }
timeout -k 10 300 python -u bench.py --workload nucleon > "$OUT/bench_nucleon.json" 2> "$OUT/bench_nucleon.err"
rc=$?; tail -c 800 "$OUT/bench_nucleon.json"; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench_nucleon.err"; exit $rc; }
timeout -k 10 300 python -u bench.py --workload stress > "$OUT/bench_stress.json" 2> "$OUT/bench_stress.err"
rc=$?; tail -c 800 "$OUT/bench_stress.json"; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench_stress.err"; exit $rc; }
MCMC_BENCH_BACKEND=gloo timeout -k 10 300 python -u bench.py --gpus 2 --workload stress --chains 524288 --steps 3 --warmup 1 > "$OUT/bench_stress_n2_gloo.json" 2> "$OUT/bench_stress_n2_gloo.err"
rc=$?; tail -c 600 "$OUT/bench_stress_n2_gloo.json"; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench_stress_n2_gloo.err"; exit $rc; }
echo ALL_OK
