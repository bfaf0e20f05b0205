#!/bin/bash
# Round 3, session W: non-temporal loads for the streaming inputs of the GEMM epilogues (residual, dres, xhat, u) -- -DMBX_LD_EPI=1.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
NTP=proj,fc2,lnb_qkv,lnb_fc1,dX_fc2
for v in base ldnt; do
  unset MBX_LIB; [ $v != base ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 200 python tools/gemm_bench.py --only $NTP > gpurun_out/r03w_gemm_$v.txt 2>&1
  echo "== $v"; grep "^nt\|rror" gpurun_out/r03w_gemm_$v.txt | cut -c1-110
done
for v in base ldnt; do
  unset MBX_LIB; [ $v != base ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03w_bench_$v.json 2> gpurun_out/r03w_bench_$v.log
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03w_bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])
except Exception as e:
    print('$v', 'failed', e); print(open('gpurun_out/r03w_bench_$v.log').read()[-800:])
PY
done
