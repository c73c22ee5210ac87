#!/bin/bash
# tools/gpu_valu_calibration.sh -- on the GPU box: the SQ counter passes behind profiles/r2_valu_busy_calibrated.json.
# Targets: the pure-instruction loops of tools/ubench_cal.hip (100 % VALU issue by construction) and the four kernels
# of the path (chain, chain with 49 lags, stress, S_AB), each under the same counter sets, one rocprofv3 run per pass
# (no multiplexing), the program itself after `--`.  Output: gpurun_out/valu_cal/<target>_<pass>/ (CSV).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/valu_cal
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PASS_A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE GRBM_COUNT"
PASS_B="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE"
PASS_C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"

run() { # name, rocprof args..., then "--" and the program
    local name=$1; shift
    echo "== $name"
    timeout -s KILL 240 rocprofv3 --output-format csv -d "$OUT/$name" -o "$name" "$@" > "$OUT/$name.out" 2> "$OUT/$name.err"
    local rc=$?
    echo "rc=$rc"; tail -n 1 "$OUT/$name.out" | cut -c1-300
    return $rc
}

target() { # tag, needs_lds_pass, program...
    local tag=$1 lds=$2; shift 2
    run ${tag}_trace --kernel-trace --stats -- "$@" || exit 1
    run ${tag}_a --pmc $PASS_A -- "$@" || exit 1
    run ${tag}_b --pmc $PASS_B -- "$@" || exit 1
    if [ "$lds" = 1 ]; then run ${tag}_c --pmc $PASS_C -- "$@" || exit 1; fi
}

BENCH_FLAGS="--steps 3 --warmup 1 --no-cpu-baseline --no-ess --no-secondary --no-clock"
target ubench 0 $ROOT/tools/bin/ubench_cal
target chain 0 python3 $ROOT/bench.py $BENCH_FLAGS
target lag49 0 python3 $ROOT/bench.py $BENCH_FLAGS --lags 49 --chains 4194304
target stress 1 python3 $ROOT/tools/bench_mvn.py --reps 2
target sab 1 python3 $ROOT/tools/bench_xs.py --which c4 --reps 2 --rule nodiv --bs 7 --inel-bs 0
find "$OUT" -name "*.csv" | wc -l
echo ALL_OK
