#!/bin/bash
# Round 3, session F: LayerNorm folding -- all new kernel tests, model fixtures, kernel trace of the folded step.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fold.py -q -p no:cacheprovider > gpurun_out/r03f_pytest_fold.log 2>&1
echo "pytest fold exit $?" > gpurun_out/r03f_summary.txt
timeout 600 python -m pytest tests/test_gpu_model.py -q -p no:cacheprovider -k "tiny_golden or baseline_shape or full_size or recompute_mode" > gpurun_out/r03f_pytest_model.log 2>&1
echo "pytest model exit $?" >> gpurun_out/r03f_summary.txt
cd /tmp; rm -rf /tmp/kt
( cd $GRAFT_REPO_ROOT && MBX_DUAL_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r03f_prof.log 2>&1 )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > gpurun_out/r03f_kernel_stats.txt 2>&1
cat gpurun_out/r03f_summary.txt; tail -30 gpurun_out/r03f_pytest_fold.log | cut -c1-600; tail -12 gpurun_out/r03f_pytest_model.log | cut -c1-600; head -30 gpurun_out/r03f_kernel_stats.txt | cut -c1-200
