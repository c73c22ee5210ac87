#!/bin/bash
# tools/gpu_trace_c4.sh -- on the GPU box: rocprofv3 --kernel-trace --stats of the two passes of configuration C4 (tools/run_c4.sh)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/c4trace
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
COMMON="isotopeA=PB-208 isotopeB=PB-208 pdf=r2ws:6.62:0.546 sigmaNN=6.76 beta=0.2 totalBlocks=128 threadsPerBlock=128 samplesForMix=8192 b0=0.05 db=0.1 ndbs=200 outDir=$OUT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p1" -o p1 -- "$ROOT/mcmc_amd/host/xsection" $COMMON \
    samplesPerThread=32 totalRuns=2 blackDisk=1 integrateRMS=1 outFile=p1 > "$OUT/p1.txt" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p2" -o p2 -- "$ROOT/mcmc_amd/host/xsection" $COMMON \
    samplesPerThread=64 totalRuns=8 inelOnly=1 rule=cross randomStart=1 integrateRMS=0 outFile=p2 > "$OUT/p2.txt" 2>&1 || exit 2
grep -h "scan\|sigma" "$OUT/p1.txt" "$OUT/p2.txt"
for p in p1 p2; do echo "== $p"; find "$OUT/$p" -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-200; done
