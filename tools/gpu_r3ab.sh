#!/bin/bash
# Round 3, session AB: one-wave attention backward with delta = rowsum(P o dP) (O no longer read): parity, timing, step.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fold.py tests/test_gpu_model.py -x -q > gpurun_out/r03ab_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r03ab_pytest.log
timeout 200 python tools/attn_bench.py > gpurun_out/r03ab_attn.txt 2>&1; grep "fwd\|bwd" gpurun_out/r03ab_attn.txt
timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03ab_bench.json 2> gpurun_out/r03ab_bench.log
python - <<PY
import json
d = json.loads(open('gpurun_out/r03ab_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['kernel_breakdown_ms'].items() if 'attn' in k})
PY
