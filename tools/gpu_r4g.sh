#!/bin/bash
mkdir -p gpurun_out
( python tools/mlp_time.py 256 512; python tools/mlp_time.py 64 512
  for v in stg0 stg1 stg2 stg5; do MBX_LIB=tools/variants/libmbx_$v.so python tools/mlp_time.py 256 512; MBX_LIB=tools/variants/libmbx_$v.so python tools/mlp_time.py 64 512; done ) 2>&1 | grep -v amdgpu | tee gpurun_out/r4g_stagger.txt
