#!/bin/bash
mkdir -p gpurun_out
( python tools/mlp_time.py 256 512
  for v in pf3 pf6 pf7 gsc mlpdbg64 mlpdbg72; do MBX_LIB=tools/variants/libmbx_$v.so python tools/mlp_time.py 256 512; done ) 2>&1 | grep -v amdgpu | tee gpurun_out/r4f_ablate.txt
