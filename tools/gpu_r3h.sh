#!/bin/bash
# Round 3, session H: full GPU suite on the folded path, default bench line (with extras), round profiles (kernel trace + PMC passes).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > gpurun_out/r03h_pytest_gpu.log 2>&1
echo "pytest exit $?" > gpurun_out/r03h_summary.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03h_bench.json 2> gpurun_out/r03h_bench.log
echo "bench exit $?" >> gpurun_out/r03h_summary.txt
TAG=r03h bash tools/gpu_profiles.sh > gpurun_out/r03h_profiles.log 2>&1
cat gpurun_out/r03h_summary.txt; tail -14 gpurun_out/r03h_pytest_gpu.log | cut -c1-300; tail -4 gpurun_out/r03h_bench.log; cut -c1-1500 gpurun_out/r03h_bench.json; cat gpurun_out/r03h_pmc_bench.txt | cut -c1-200
