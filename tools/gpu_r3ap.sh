#!/bin/bash
# Round 3, session AP: timing probe (results wrong on purpose): the 256x128 NT kernel reading its WEIGHT operand as if packed k-tile-major
# (MBX_DBG=64: whole-line LDS-DMA requests for W) -- what would pre-packed weights buy?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp MBX_LIB=tools/variants/libmbx_diag.so
for d in 0 64 0 64; do
  MBX_DBG=$d timeout 200 python tools/gemm_bench.py --only proj,fc2,lnb_qkv,lnb_fc1 --check 0 > gpurun_out/r03ap_dbg$d.txt 2>&1
  echo "== dbg $d"; grep "^nt\|rror" gpurun_out/r03ap_dbg$d.txt | cut -c1-75
done
