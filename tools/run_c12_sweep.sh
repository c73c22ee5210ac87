#!/bin/bash
# The reference's C-12 + C-12 sweep (MacroGenerator.ipynb cell 2: c = linspace(1.5, 1.7, 20), out<k>.root) with 64 runs
# per impact parameter: prints k, c, sigma_tot, sigma_el, sigma_rxn for comparison with the values recorded in
# XSectionAnalysis.ipynb cell 5 (Total 1331.2473, Elastic 476.0962, Reaction 855.1511 mb for "out1.root").
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/c12
mkdir -p $OUT
cd $ROOT
for k in $(seq 0 19); do
    c=$(python3 -c "print(repr(1.5 + 0.2 * $k / 19))")
    ./mcmc_amd/host/xsection isotopeA=C-12 isotopeB=C-12 pdf=r2ws:$c:0.54 sigmaNN=4.3 beta=0.2 stepsize=2.2 rule=recip totalBlocks=64 \
        threadsPerBlock=128 samplesPerThread=64 samplesForMix=8192 b0=0.05 db=0.1 ndbs=200 integrateRMS=0 totalRuns=64 \
        outDir=$OUT outFile=sweep$k > $OUT/sweep$k.txt
    echo "$k $c $(grep sigma_tot $OUT/sweep$k.txt | awk '{print $3, $5}') $(grep sigma_el $OUT/sweep$k.txt | awk '{print $3, $5}') $(grep sigma_rxn $OUT/sweep$k.txt | awk '{print $3, $5}') $(sed -n 2p $OUT/sweep$k.S_AB.csv)"
done
