#!/bin/bash
# Round 3, session AE: the default bench line (with the extras and the CPU baseline) on the final kernels; smoke().
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.log; echo "bench exit $?"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03g_smoke.log 2>&1; tail -2 gpurun_out/r03g_smoke.log
tail -4 gpurun_out/r03g_bench.log; cut -c1-400 gpurun_out/r03g_bench.json
