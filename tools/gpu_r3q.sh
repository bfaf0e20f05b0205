#!/bin/bash
# Round 3, session Q: (1) non-temporal output stores of the 256x256 kernel's bf16 epilogues, re-measured with correct data
# (session P's asm store lacked the store-data hazard nops); (2) where the 256x128 NT kernel's time goes: ablations of the
# diagnostic build (MBX_DBG 1 = no MFMA block, 2 = no LDS-DMA, 4 = no epilogue, 8 = every tile reads the same, L2-resident A panel)
# and cycle stamps of one workgroup.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
PPS=qkv,fc1,dX_qkv,dX_fc1,st1024
timeout 200 python tools/gemm_bench.py --only $PPS > gpurun_out/r03q_gemm_nt.txt 2>&1
MBX_LIB=tools/variants/libmbx_st_plain.so timeout 200 python tools/gemm_bench.py --only $PPS > gpurun_out/r03q_gemm_plain.txt 2>&1
for v in nt plain; do echo "== stores $v"; grep "^nt" gpurun_out/r03q_gemm_$v.txt | cut -c1-110; done
NTP=proj,fc2,lnb_qkv,lnb_fc1
for d in 0 1 2 4 8 12 5 6; do
  MBX_LIB=tools/variants/libmbx_diag.so MBX_DBG=$d timeout 200 python tools/gemm_bench.py --only $NTP --check 0 > gpurun_out/r03q_ntp_dbg$d.txt 2>&1
  echo "== dbg $d"; grep "^nt" gpurun_out/r03q_ntp_dbg$d.txt | cut -c1-75
done
for m in "512 1536 lnbwd" "512 1024 resid"; do
  MBX_LIB=tools/variants/libmbx_trace.so timeout 100 python tools/nt_trace.py $m > "gpurun_out/r03q_trace_$(echo $m | tr ' ' _).txt" 2>&1
  tail -12 "gpurun_out/r03q_trace_$(echo $m | tr ' ' _).txt"
done
for v in base st_plain; do
  unset MBX_LIB; [ $v != base ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03q_bench_$v.json 2> gpurun_out/r03q_bench_$v.log
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03q_bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])
except Exception as e:
    print('$v', 'failed', e); print(open('gpurun_out/r03q_bench_$v.log').read()[-800:])
PY
done
unset MBX_LIB
timeout 600 python -m pytest tests -m gpu -x -q -k "gemm or nt or model" > gpurun_out/r03q_pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/r03q_pytest.log
