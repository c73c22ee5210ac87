#!/bin/bash
# tools/gpu_writeback_pmc.sh -- on the GPU box: where the chain kernel's sample write-back loses its overlap.  PMC passes (one
# run each, no multiplexing) of tools/bench_writeback.py 256 {none,rows,pairs}: shader cycles and clock (GRBM_GUI_ACTIVE over the
# dispatch's duration), issue / wait cycles, vector-memory instructions in flight, and the write path TCP -> TCC -> fabric.
# Writes gpurun_out/wb_pmc/<case>_<pass>/ (CSV); tools/summarize_writeback_pmc.py condenses them.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/wb_pmc
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
SWEEPS=${1:-256}
run() { # case, pass, counters...
    local tag=$1 pass=$2; shift 2
    local name=${tag}_$pass
    echo "== $name"
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 $ROOT/tools/bench_writeback.py $SWEEPS $tag > "$OUT/$name.out" 2> "$OUT/$name.err"
    local rc=$?
    echo "rc=$rc"; tail -n 1 "$OUT/$name.out"
    return $rc
}
for c in none rows pairs; do
    run $c sq GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU || exit 1
    [ $c = none ] && continue
    run $c tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum || exit 1
    run $c tcc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum || exit 1
done
echo ALL_OK
