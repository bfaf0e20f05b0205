#!/bin/bash
# Round 3, session AL: s_setprio around the MFMA phases (prio1..3) / the fragment-read phase (prior1) of the ping-pong GEMMs.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
S=qkv,fc1,dX_qkv,dX_fc2,dW_qkv,dW_fc1,dW_proj
for v in base prio1 prio2 prio3 prior1; do
  unset MBX_LIB; [ $v != base ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 200 python tools/gemm_bench.py --only $S --check 0 > gpurun_out/r03al_$v.txt 2>&1
  echo "== $v"; grep "^nt\|^tn\|rror" gpurun_out/r03al_$v.txt | cut -c1-75
done
