#!/bin/bash
# Round 3, session M (final artefacts): smoke(), the driver's default bench command (with cpu_baseline), round profiles.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python __graft_entry__.py smoke > gpurun_out/r03m_smoke.log 2>&1
echo "smoke exit $?" > gpurun_out/r03m_summary.txt
timeout 900 python bench.py > gpurun_out/r03m_bench.json 2> gpurun_out/r03m_bench.log
echo "bench exit $?" >> gpurun_out/r03m_summary.txt
TAG=r03m bash tools/gpu_profiles.sh > gpurun_out/r03m_profiles.log 2>&1
cat gpurun_out/r03m_summary.txt; tail -5 gpurun_out/r03m_smoke.log; tail -3 gpurun_out/r03m_bench.log; cut -c1-400 gpurun_out/r03m_bench.json; head -12 gpurun_out/r03m_pmc_bench.txt | cut -c1-160
