#!/bin/bash
# Round 3, session S: producer-wave variant of the 256x128 NT kernel (-DMBX_NTP_PW=1) against the product kernel.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
NTP=proj,fc2,lnb_qkv,lnb_fc1
timeout 200 python tools/gemm_bench.py --only $NTP > gpurun_out/r03s_base.txt 2>&1
MBX_LIB=tools/variants/libmbx_pw.so timeout 200 python tools/gemm_bench.py --only $NTP > gpurun_out/r03s_pw.txt 2>&1
for v in base pw; do echo "== $v"; grep "^nt\|rror" gpurun_out/r03s_$v.txt | cut -c1-110; done
for d in 4 5 6; do
  MBX_LIB=tools/variants/libmbx_pwdiag.so MBX_DBG=$d timeout 200 python tools/gemm_bench.py --only $NTP --check 0 > gpurun_out/r03s_pw_dbg$d.txt 2>&1
  echo "== pw dbg $d"; grep "^nt\|rror" gpurun_out/r03s_pw_dbg$d.txt | cut -c1-75
done
