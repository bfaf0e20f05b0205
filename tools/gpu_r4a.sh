#!/bin/bash
# round 4, session A: first run of the fused MLP forward + raw-operand GEMMs (parity, then timing)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rawln.py -x -q 2>&1 | tail -25 > gpurun_out/r4a_tests.txt
cat gpurun_out/r4a_tests.txt
timeout 300 python tools/mlp_bench.py 64 256 2>&1 | tee gpurun_out/r4a_mlp_bench.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_nt" 2>&1 | tail -5 | tee gpurun_out/r4a_gemm_tests.txt
