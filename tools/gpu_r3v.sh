#!/bin/bash
# Round 3, session V: where the fused residual-GEMM + LayerNorm loses its gain: hand-shake only (MBX_DBG=16), neither hand-shake nor tail (32).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp MBX_LIB=tools/variants/libmbx_diag.so
for d in 0 16 32; do
  MBX_DBG=$d timeout 200 python tools/gemm_bench.py --only proj,proj_ln,fc2,fc2_ln --check 0 > gpurun_out/r03v_dbg$d.txt 2>&1
  echo "== dbg $d"; grep "^nt\|rror" gpurun_out/r03v_dbg$d.txt | cut -c1-75
done
