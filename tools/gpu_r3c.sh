#!/bin/bash
# Round 3, session C: full GPU suite, default bench line, config 4 / 5 workloads.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > gpurun_out/r03c_pytest_gpu.log 2>&1
echo "pytest exit $?" > gpurun_out/r03c_summary.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.log
echo "bench exit $?" >> gpurun_out/r03c_summary.txt
timeout 300 python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline --workload pretrain > gpurun_out/r03c_bench_pretrain.json 2> gpurun_out/r03c_bench_pretrain.log
echo "bench pretrain exit $?" >> gpurun_out/r03c_summary.txt
timeout 300 python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline --workload action > gpurun_out/r03c_bench_action.json 2> gpurun_out/r03c_bench_action.log
echo "bench action exit $?" >> gpurun_out/r03c_summary.txt
cat gpurun_out/r03c_summary.txt; tail -12 gpurun_out/r03c_pytest_gpu.log; tail -2 gpurun_out/r03c_bench*.log; cut -c1-900 gpurun_out/r03c_bench.json gpurun_out/r03c_bench_pretrain.json gpurun_out/r03c_bench_action.json
