#!/bin/bash
# round 4, session D: where the fused MLP kernel's time goes (ablation builds), and the no-grad model numbers
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "no_grad" 2>&1 | tail -5 | cut -c1-300 | tee gpurun_out/r4d_tests.txt
( python tools/mlp_time.py 256 512
  for v in 1 2 4 8 16 32 9 11 3; do MBX_LIB=tools/variants/libmbx_mlpdbg$v.so python tools/mlp_time.py 256 512; done ) 2>&1 | grep -v amdgpu | tee gpurun_out/r4d_ablate.txt
timeout 600 python bench.py --block 2>&1 | tail -1 | cut -c1-600 | tee gpurun_out/r4d_block.txt
