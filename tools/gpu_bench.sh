#!/bin/bash
# bench + rocprofv3 kernel trace on the GPU box; outputs under gpurun_out/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
nproc > gpurun_out/host.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/host.txt 2>&1; lscpu | head -20 >> gpurun_out/host.txt; free -g >> gpurun_out/host.txt
B1=${B1:-8}; B2=${B2:-64}
timeout 240 python bench.py --steps 3 --warmup 2 --batch $B1 --no-cpu-baseline > gpurun_out/bench_b$B1.log 2>&1; echo "bench B=$B1 exit $?" >> gpurun_out/summary.txt
timeout 420 python bench.py --steps 5 --warmup 2 --batch $B2 > gpurun_out/bench_b$B2.log 2>&1; echo "bench B=$B2 exit $?" >> gpurun_out/summary.txt
export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01 -- python bench.py --steps 2 --warmup 1 --batch $B2 --no-cpu-baseline > gpurun_out/prof.log 2>&1; echo "rocprof exit $?" >> gpurun_out/summary.txt
find gpurun_out/prof -name "*stats*" | head >> gpurun_out/summary.txt
cat gpurun_out/summary.txt; tail -2 gpurun_out/bench_b$B1.log; tail -2 gpurun_out/bench_b$B2.log
