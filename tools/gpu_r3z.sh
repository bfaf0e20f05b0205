#!/bin/bash
# Round 3, session Z: attention backward prologues with ONE memory round trip (statistics' loads ahead of / together with the tile fill).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fold.py -x -q -k "attn or attention" > gpurun_out/r03z_pytest_attn.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03z_pytest_attn.log
for v in new attn_old; do
  unset MBX_LIB; [ $v != new ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 200 python tools/attn_bench.py > gpurun_out/r03z_attn_$v.txt 2>&1
  echo "== $v"; tail -8 gpurun_out/r03z_attn_$v.txt
done
