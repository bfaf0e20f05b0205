#!/bin/bash
# Round 3, session AN: head backward reading the representation once (parity, time), plus the whole-model gates.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "head or fixture or reset_classifier or oracle" > gpurun_out/r03an_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03an_pytest.log
timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03an_bench.json 2> gpurun_out/r03an_bench.log
python - <<PY
import json
d = json.loads(open('gpurun_out/r03an_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_breakdown_ms'].get('head_bwd'))
PY
