#!/bin/bash
# Round 3, session J: in-kernel attention-probability dropout (kernel + model tests), regression check of the step.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_library_abi.py -q -p no:cacheprovider -k "attention or attn or dropout or abi or device_code" > gpurun_out/r03j_pytest.log 2>&1
echo "pytest exit $?" > gpurun_out/r03j_summary.txt
timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03j_bench.json 2> gpurun_out/r03j_bench.log
cat gpurun_out/r03j_summary.txt; tail -12 gpurun_out/r03j_pytest.log | cut -c1-400; cut -c1-300 gpurun_out/r03j_bench.json
