#!/bin/bash
# Round 3, session P: full GPU suite on the final weight-gradient kernel; A/B of the epilogue store policy (sc1 / nt) and of a
# one-time start stagger of the second workgroup per CU in the 256x128 NT kernel (residual / LayerNorm-backward epilogues).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
NT=qkv,proj,fc1,fc2,dX_fc2,lnb_qkv,lnb_fc1
timeout 200 python tools/gemm_bench.py --only $NT > gpurun_out/r03p_gemm_base.txt 2>&1
for v in st_sc1 st_nt stag2 stag4 stag8 stag4m1; do
  MBX_LIB=tools/variants/libmbx_$v.so timeout 200 python tools/gemm_bench.py --only $NT > gpurun_out/r03p_gemm_$v.txt 2>&1
done
for v in base st_sc1 st_nt stag2 stag4 stag8 stag4m1; do echo "== $v"; grep "^nt" gpurun_out/r03p_gemm_$v.txt | cut -c1-75; done
for v in base st_sc1 stag4; do
  unset MBX_LIB; [ $v != base ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03p_bench_$v.json 2> gpurun_out/r03p_bench_$v.log
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03p_bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])
except Exception as e:
    print('$v', 'failed', e); print(open('gpurun_out/r03p_bench_$v.log').read()[-800:])
PY
done
unset MBX_LIB
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03p_pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/r03p_pytest_gpu.log
