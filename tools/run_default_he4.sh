#!/bin/bash
# tools/run_default_he4.sh -- on the GPU box: the reference's DEFAULT configuration (main.cu:38-82: He-4 + He-4, 256 x 128 threads, 256 samples,
# 16 runs, 200 impact parameters) through the xsection driver, plain and under rocprofv3 --kernel-trace --stats: wall time of
# the scan against the summed kernel time (are the 3200 small launches launch-bound?).
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/defrun
timeout -k 10 120 ./mcmc_amd/host/xsection outDir=gpurun_out/defrun outFile=he4 blackDisk=0 seed=1 > gpurun_out/defrun/stdout.txt 2>&1 || exit 1
grep -v "^Nucleon" gpurun_out/defrun/stdout.txt
export TMPDIR=/tmp
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/defrun/prof -o he4 -- $GRAFT_REPO_ROOT/mcmc_amd/host/xsection outDir=$GRAFT_REPO_ROOT/gpurun_out/defrun outFile=he4p blackDisk=0 seed=1 > $GRAFT_REPO_ROOT/gpurun_out/defrun/stdout_prof.txt 2>&1 || exit 2
grep -v "^Nucleon" $GRAFT_REPO_ROOT/gpurun_out/defrun/stdout_prof.txt | tail -6
find $GRAFT_REPO_ROOT/gpurun_out/defrun/prof -name "*kernel_stats.csv" | head -1 | xargs head -12
