#!/bin/bash
# Round 3, session R: is the 256x128 NT kernel's k-loop bound by bytes in flight (latency) or by the request rate of the CU's
# vector-memory path?  Ablations of the diagnostic build with ONE workgroup per CU (LDS padding) against two.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp MBX_LIB=tools/variants/libmbx_diag.so
NTP=fc2,lnb_qkv
for cfg in "0 0" "0 16" "4 0" "4 16" "5 0" "5 16" "13 0" "13 16" "6 0" "6 16" "2 0" "2 16" "1 0" "1 16"; do
  set -- $cfg
  MBX_DBG=$1 MBX_NTP_LDS_PAD=$2 timeout 200 python tools/gemm_bench.py --only $NTP --check 0 > gpurun_out/r03r_dbg$1_pad$2.txt 2>&1
  echo "== dbg $1 pad $2"; grep "^nt" gpurun_out/r03r_dbg$1_pad$2.txt | cut -c1-75
done
