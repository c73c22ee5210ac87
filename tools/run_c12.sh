#!/bin/bash
# The one xsection run whose OUTPUT the reference records: xsection/tools/XSectionAnalysis.ipynb cells 1 and 5 show the
# S_AB table head and sigma_tot / sigma_el / sigma_rxn of ../data/out1.root, which MacroGenerator.ipynb cells 1-2
# generate as C-12 + C-12, R2WoodsSaxon c = linspace(1.5, 1.7, 20)[1] = 1.5105263, a = 0.54, b = 0.05 + 0.1 k (k < 200),
# totalRuns 8, totalBlocks 64, threadsPerBlock 128, samplesPerThread 64, samplesForMix 2^13 (sigmaNN 4.3, beta 0.2,
# width 2.2: IntegrateSAB.cuh:10-11, main.cu:149).  Pass 1 repeats those parameters, pass 2 uses 32x the runs.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/c12
mkdir -p $OUT
cd $ROOT
COMMON="isotopeA=C-12 isotopeB=C-12 pdf=r2ws:1.5105263158:0.54 sigmaNN=4.3 beta=0.2 stepsize=2.2 rule=recip totalBlocks=64 threadsPerBlock=128 samplesPerThread=64 samplesForMix=8192 b0=0.05 db=0.1 ndbs=200 integrateRMS=1 outDir=$OUT"
./mcmc_amd/host/xsection $COMMON totalRuns=8 outFile=c12_ref_params > $OUT/stdout_ref_params.txt
grep -v "^Nucleon" $OUT/stdout_ref_params.txt
head -6 $OUT/c12_ref_params.S_AB.csv
./mcmc_amd/host/xsection $COMMON totalRuns=256 outFile=c12_256runs > $OUT/stdout_256runs.txt
grep "sigma_\|S_AB scan" $OUT/stdout_256runs.txt
head -6 $OUT/c12_256runs.S_AB.csv
