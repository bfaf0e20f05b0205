#!/bin/bash
# round 4, session C: the no-grad sequencing in the model -- parity on fixtures, then forward-only / Block numbers
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "no_grad or infer_wild or hipgraph or config0 or non_contiguous or native" 2>&1 | tail -15 | tee gpurun_out/r4c_tests.txt
timeout 600 python bench.py --block 2>&1 | tail -3 | tee gpurun_out/r4c_block.txt
MBX_RAWLN=0 timeout 600 python bench.py --block 2>&1 | tail -1 | tee gpurun_out/r4c_block_plain.txt
timeout 600 python tools/fwd_bench.py 2>&1 | tail -8 | tee gpurun_out/r4c_fwd.txt
MBX_RAWLN=0 timeout 600 python tools/fwd_bench.py 2>&1 | tail -8 | tee gpurun_out/r4c_fwd_plain.txt
