#!/bin/bash
# Round 3, session A: full GPU suite (new parity tests), default bench line, dual-stream kernel trace for the overlap analysis.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 -x > gpurun_out/r03a_pytest_gpu.log 2>&1
echo "pytest exit $?" > gpurun_out/r03a_summary.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.log
echo "bench exit $?" >> gpurun_out/r03a_summary.txt
cd /tmp; rm -rf /tmp/kt2
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt2 -o kt -- python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r03a_prof_dual.log 2>&1 )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/kt2 -name "*.db" | head -1)
python tools/rocpd_overlap.py $DB 1800 > gpurun_out/r03a_overlap_dual.txt 2>&1
cat gpurun_out/r03a_summary.txt; tail -6 gpurun_out/r03a_pytest_gpu.log; tail -3 gpurun_out/r03a_bench.log; cat gpurun_out/r03a_overlap_dual.txt
