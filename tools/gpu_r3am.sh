#!/bin/bash
# Round 3, session AM: sixteen-wave weight-gradient GEMM (-DMBX_TN_W16=1: two waves per SIMD in the transpose-read phase) against the eight-wave kernel.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
TN=dW_qkv,dW_proj,dW_fc1,dW_fc2
timeout 200 python tools/gemm_bench.py --only $TN > gpurun_out/r03am_base.txt 2>&1; echo "== base"; grep "^tn\|rror" gpurun_out/r03am_base.txt | cut -c1-100
export MBX_LIB=tools/variants/libmbx_tnw16.so
timeout 200 python tools/gemm_bench.py --only $TN > gpurun_out/r03am_w16.txt 2>&1; echo "== w16"; grep "^tn\|rror" gpurun_out/r03am_w16.txt | cut -c1-100
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "tn" > gpurun_out/r03am_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03am_pytest.log
