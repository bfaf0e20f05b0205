#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/mlp_debug.py 512 2>&1 | tee gpurun_out/r4b_debug512.txt | head -150
timeout 300 python tools/mlp_debug.py 256 2>&1 | tee gpurun_out/r4b_debug256.txt | head -100
