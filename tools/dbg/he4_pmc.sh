#!/bin/bash
# development: PMC pass of the He-4 + He-4 S_AB kernel at the reference's default launch size
set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/he4pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/p1 -o p1 -- $GRAFT_REPO_ROOT/mcmc_amd/host/xsection ndbs=3 totalRuns=2 integrateRMS=0 outDir=$OUT outFile=x > $OUT/p1.txt 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_IFETCH --output-format csv -d $OUT/p2 -o p2 -- $GRAFT_REPO_ROOT/mcmc_amd/host/xsection ndbs=3 totalRuns=2 integrateRMS=0 outDir=$OUT outFile=x > $OUT/p2.txt 2>&1 || exit 2
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    f = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(p, "no counter file", glob.glob(f"{out}/{p}/**/*", recursive=True)[:5]); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if "sab_kernel" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print(p, k, v / max(n, 1), n)
PY
