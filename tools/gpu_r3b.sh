#!/bin/bash
# Round 3, session B: full GPU suite, bench workloads (config 4 / 5), dual-stream pair overlap.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 > gpurun_out/r03b_pytest_gpu.log 2>&1
echo "pytest exit $?" > gpurun_out/r03b_summary.txt
timeout 300 python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline --workload pretrain > gpurun_out/r03b_bench_pretrain.json 2> gpurun_out/r03b_bench_pretrain.log
echo "bench pretrain exit $?" >> gpurun_out/r03b_summary.txt
timeout 300 python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline --workload action > gpurun_out/r03b_bench_action.json 2> gpurun_out/r03b_bench_action.log
echo "bench action exit $?" >> gpurun_out/r03b_summary.txt
cd /tmp; rm -rf /tmp/kt2
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt2 -o kt -- python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r03b_prof_dual.log 2>&1 )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/kt2 -name "*.db" | head -1)
python tools/rocpd_overlap.py $DB 1800 > gpurun_out/r03b_overlap_dual.txt 2>&1
cat gpurun_out/r03b_summary.txt; tail -8 gpurun_out/r03b_pytest_gpu.log; tail -2 gpurun_out/r03b_bench_pretrain.log gpurun_out/r03b_bench_action.log; cut -c1-600 gpurun_out/r03b_bench_pretrain.json gpurun_out/r03b_bench_action.json; cat gpurun_out/r03b_overlap_dual.txt
