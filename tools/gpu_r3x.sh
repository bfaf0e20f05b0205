#!/bin/bash
# Round 3, session X: the 256x128 NT kernel with BK = 64 (whole-line LDS-DMA requests, one workgroup per CU) against BK = 32.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
NTP=proj,fc2,lnb_qkv,lnb_fc1,proj_ln
for v in base bk64; do
  unset MBX_LIB; [ $v != base ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 200 python tools/gemm_bench.py --only $NTP > gpurun_out/r03x_gemm_$v.txt 2>&1
  echo "== $v"; grep "^nt\|rror" gpurun_out/r03x_gemm_$v.txt | cut -c1-110
done
export MBX_LIB=tools/variants/libmbx_bk64diag.so
for d in 4 5 6; do
  MBX_DBG=$d timeout 200 python tools/gemm_bench.py --only proj,fc2,lnb_qkv,lnb_fc1 --check 0 > gpurun_out/r03x_bk64_dbg$d.txt 2>&1
  echo "== bk64 dbg $d"; grep "^nt\|rror" gpurun_out/r03x_bk64_dbg$d.txt | cut -c1-75
done
unset MBX_LIB
for v in base bk64; do
  unset MBX_LIB; [ $v != base ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03x_bench_$v.json 2> gpurun_out/r03x_bench_$v.log
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03x_bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])
except Exception as e:
    print('$v', 'failed', e); print(open('gpurun_out/r03x_bench_$v.log').read()[-800:])
PY
done
