#!/bin/bash
# Round 3, session Y: L2 prefetch of the epilogue inputs through the scalar path (s_atc_probe / s_load) a few k-tiles ahead.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
NTP=proj,fc2,lnb_qkv,lnb_fc1
for v in base spfa3 spfa8 spfl3 spfl8; do
  unset MBX_LIB; [ $v != base ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 200 python tools/gemm_bench.py --only $NTP > gpurun_out/r03y_gemm_$v.txt 2>&1
  echo "== $v"; grep "^nt\|rror" gpurun_out/r03y_gemm_$v.txt | cut -c1-110
done
