#!/bin/bash
# Round 3, session AJ: full GPU suite on the final tree (no -x: every failure in one run).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03h_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -6 gpurun_out/r03h_pytest_gpu.log
