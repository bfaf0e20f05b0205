#!/bin/bash
# Round 3, session AI: closing run on the final tree: full GPU suite, smoke, round profiles (kernel trace + PMC), GEMM / attention tables.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03h_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03h_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03h_smoke.log 2>&1; tail -1 gpurun_out/r03h_smoke.log
TAG=r03h bash tools/gpu_profiles.sh > gpurun_out/r03h_profiles.log 2>&1
timeout 300 python tools/gemm_bench.py > gpurun_out/r03h_gemm_shapes.txt 2>&1
timeout 100 python tools/attn_bench.py > gpurun_out/r03h_attn.txt 2>&1
head -12 gpurun_out/r03h_pmc_bench.txt | cut -c1-170
