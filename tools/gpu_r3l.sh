#!/bin/bash
# Round 3, session L: full GPU suite after the attention changes (dropout variants, row-dot vectors in the dead dO tile), step + trace.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/r03l_pytest_gpu.log 2>&1
echo "pytest exit $?" > gpurun_out/r03l_summary.txt
timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03l_bench.json 2> gpurun_out/r03l_bench.log
cd /tmp; rm -rf /tmp/kt
( cd $GRAFT_REPO_ROOT && MBX_DUAL_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r03l_prof.log 2>&1 )
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) > gpurun_out/r03l_kernel_stats.txt 2>&1
cat gpurun_out/r03l_summary.txt; tail -4 gpurun_out/r03l_pytest_gpu.log | cut -c1-300; cut -c1-260 gpurun_out/r03l_bench.json; grep "attn" gpurun_out/r03l_kernel_stats.txt | cut -c1-130
