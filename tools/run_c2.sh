#!/bin/bash
# BASELINE config 1 (C2): acceptance + integrated autocorrelation time vs proposal width, 3 coordinates of
# r^2 WS(2.0, 0.54), 2^20 chains, 1e3 warm-up + 1e4 measured steps, widths 0.01..5.00, lags 1..49.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/c2
mkdir -p $OUT
cd $ROOT
./mcmc_amd/host/stepsize --chains 1048576 --coords 3 --iterations 10000 --warmup 1000 \
    --precision 32 --rule cross --math fast --outdir $OUT > $OUT/stdout.txt
tail -2 $OUT/stdout.txt
python3 - <<PY
import numpy as np
e=np.loadtxt("$OUT/efficiency.txt"); s=np.loadtxt("$OUT/statistics.txt")
i=e[:,3].argmax(); j=s[:,2].argmin()
print("max efficiency %.4f at width %.2f; min lag-1 %.4f at width %.2f" % (e[i,3], e[i,0], s[j,2], s[j,0]))
print("acceptance @0.01 %.6f @5.00 %.6f; lag-1 @0.01 %.6f @4.96 %.6f @5.00 %.6f" % (s[0,1], s[-1,1], s[0,2], s[495,2], s[-1,2]))
PY
