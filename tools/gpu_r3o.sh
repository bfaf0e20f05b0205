#!/bin/bash
# Round 3, session O: default bench line and round profiles after the weight-gradient GEMM's bias-gradient balancing.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03o_bench.json 2> gpurun_out/r03o_bench.log
echo "bench exit $?" > gpurun_out/r03o_summary.txt
TAG=r03o bash tools/gpu_profiles.sh > gpurun_out/r03o_profiles.log 2>&1
timeout 300 python tools/gemm_bench.py > gpurun_out/r03o_gemm_shapes.txt 2>&1
cat gpurun_out/r03o_summary.txt; tail -3 gpurun_out/r03o_bench.log; cut -c1-330 gpurun_out/r03o_bench.json; head -14 gpurun_out/r03o_pmc_bench.txt | cut -c1-160; grep "^nt\|^tn" gpurun_out/r03o_gemm_shapes.txt
