set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/s1_tests.log 2>&1; rc=$?; tail -5 gpurun_out/s1_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --rule recip > gpurun_out/s1_bench_recip.json 2> gpurun_out/s1_bench_recip.err || exit 1
timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --rule cross > gpurun_out/s1_bench_cross.json 2> gpurun_out/s1_bench_cross.err || exit 1
cat gpurun_out/s1_bench_recip.json gpurun_out/s1_bench_cross.json | cut -c1-400
rocprofv3 -L > gpurun_out/s1_counters.txt 2>&1 || true
bash tools/gpu_profile.sh r1v2 --rule cross
