#!/bin/bash
# Round 3, session AF: persistent temporal attention backward (-DMBX_ATTN_BWD_PERS=1) against the fused kernel: parity, timing.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 100 python tools/attn_bench.py > gpurun_out/r03af_attn_base.txt 2>&1; echo "== base"; grep "temporal bwd" gpurun_out/r03af_attn_base.txt
export MBX_LIB=tools/variants/libmbx_pers1.so
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fold.py -x -q -k "attn or attention" > gpurun_out/r03af_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03af_pytest.log
timeout 100 python tools/attn_bench.py > gpurun_out/r03af_attn_pers.txt 2>&1; echo "== pers"; grep "temporal bwd" gpurun_out/r03af_attn_pers.txt
