#!/bin/bash
# Round 3, session AG: packed-fp32 GELU (A&S 7.1.28, no exponential) / GELU' epilogues of the 256x256 kernel against the scalar forms.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fold.py -x -q -k "gemm or gelu or dgelu" > gpurun_out/r03ag_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03ag_pytest.log
for v in new gelu_old; do
  unset MBX_LIB; [ $v != new ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 200 python tools/gemm_bench.py --only fc1,fc1_noU,dX_fc2,st1024 > gpurun_out/r03ag_gemm_$v.txt 2>&1
  echo "== $v"; grep "^nt\|rror" gpurun_out/r03ag_gemm_$v.txt | cut -c1-110
done
for v in new gelu_old; do
  unset MBX_LIB; [ $v != new ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03ag_bench_$v.json 2> gpurun_out/r03ag_bench_$v.log
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03ag_bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])
except Exception as e:
    print('$v', 'failed', e); print(open('gpurun_out/r03ag_bench_$v.log').read()[-800:])
PY
done
