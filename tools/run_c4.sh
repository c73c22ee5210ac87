#!/bin/bash
# BASELINE config 3 (C4) on one GPU: MC Glauber Pb-208 + Pb-208, sigma_NN^inel(5.02 TeV) = 67.6 mb,
# literature Woods-Saxon (c = 6.62 fm, a = 0.546 fm; not from the reference), b = 0.05 + 0.1 k, k < 200
# (MacroGenerator.ipynb cell 1).
#   pass 1: S_AB + all black-disk counters, 2^14 events x 32 samples x 2 runs per b = 2.1e8 event samples
#   pass 2: inelastic flag only (inelOnly=1), 2^14 events x 64 samples x 8 runs per b = 1.7e9 event samples
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/c4
mkdir -p $OUT
cd $ROOT
./mcmc_amd/host/xsection isotopeA=PB-208 isotopeB=PB-208 pdf=r2ws:6.62:0.546 sigmaNN=6.76 beta=0.2 \
    totalBlocks=128 threadsPerBlock=128 samplesPerThread=32 totalRuns=2 samplesForMix=8192 \
    b0=0.05 db=0.1 ndbs=200 blackDisk=1 integrateRMS=1 outDir=$OUT outFile=pbpb > $OUT/stdout.txt
grep -v "^Nucleon" $OUT/stdout.txt
grep "^Nucleon" $OUT/stdout.txt | awk '{s+=$6; n++} END{print "mean nucleon RMS", s/n, "fm over", n, "nucleons (quadrature 5.514875)"}'
./mcmc_amd/host/xsection isotopeA=PB-208 isotopeB=PB-208 pdf=r2ws:6.62:0.546 sigmaNN=6.76 beta=0.2 \
    totalBlocks=128 threadsPerBlock=128 samplesPerThread=64 totalRuns=8 samplesForMix=8192 \
    b0=0.05 db=0.1 ndbs=200 inelOnly=1 rule=cross randomStart=1 integrateRMS=0 outDir=$OUT outFile=pbpb > $OUT/stdout_inel.txt
cat $OUT/stdout_inel.txt
