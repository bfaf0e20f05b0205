#!/bin/bash
# round 4, session E: fused MLP v2 (accumulators start from residual + bias; leaner DMA / fragment addressing): parity, ablations
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rawln.py -x -q 2>&1 | tail -4 | cut -c1-300 | tee gpurun_out/r4e_tests.txt
( python tools/mlp_time.py 256 512; python tools/mlp_time.py 64 512; python tools/mlp_time.py 256 256
  for v in gsc mlpdbg1 mlpdbg8 mlpdbg9 mlpdbg16; do MBX_LIB=tools/variants/libmbx_$v.so python tools/mlp_time.py 256 512; done ) 2>&1 | grep -v amdgpu | tee gpurun_out/r4e_ablate.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "no_grad" 2>&1 | tail -3 | cut -c1-300 | tee -a gpurun_out/r4e_tests.txt
timeout 600 python bench.py --block 2>&1 | tail -1 | cut -c1-500 | tee gpurun_out/r4e_block.txt
