#!/bin/bash
# tools/gpu_store_pmc.sh -- on the GPU box: where the write-back variants lose their time.  PMC passes (one run each, no
# multiplexing) of tools/bench_mvn.py (register kernel with and without full-state write-back) and tools/bench_store.py:
# address translation (UTCL1), the SQ -> TA queues, the TCP and the L2 -> fabric write path.
# Writes gpurun_out/store_pmc/<prog>_<pass>/ (CSV).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/store_pmc
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() { # prog-tag, pass, program args..., then counters after --
    local tag=$1 pass=$2 prog=$3; shift 3
    local name=${tag}_$pass
    echo "== $name"
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 $ROOT/tools/$prog > "$OUT/$name.out" 2> "$OUT/$name.err"
    local rc=$?
    echo "rc=$rc"
    return $rc
}
for tp in "mvn bench_mvn.py" "store bench_store.py"; do
    set -- $tp
    run $1 tlb $2 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE || exit 1
    run $1 sq $2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL || exit 1
    run $1 tcp $2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_TCC_WRITE_REQ_sum || exit 1
    run $1 tcc $2 TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum || exit 1
done
echo ALL_OK
