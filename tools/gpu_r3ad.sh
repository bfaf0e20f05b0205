#!/bin/bash
# Round 3, session AD: round profiles (kernel trace + three PMC passes of the bench command), GEMM and attention tables.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
TAG=r03f bash tools/gpu_profiles.sh > gpurun_out/r03f_profiles.log 2>&1
timeout 300 python tools/gemm_bench.py > gpurun_out/r03f_gemm_shapes.txt 2>&1
timeout 100 python tools/attn_bench.py > gpurun_out/r03f_attn.txt 2>&1
head -14 gpurun_out/r03f_pmc_bench.txt | cut -c1-170; grep "^nt\|^tn\|^  " gpurun_out/r03f_gemm_shapes.txt | cut -c1-80; grep "fwd\|bwd" gpurun_out/r03f_attn.txt
