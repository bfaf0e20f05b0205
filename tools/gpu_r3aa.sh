#!/bin/bash
# Round 3, session AA: full GPU suite + bench after the one-round-trip attention prologues (A/B against the old prologues).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
for v in new attn_old; do
  unset MBX_LIB; [ $v != new ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03aa_bench_$v.json 2> gpurun_out/r03aa_bench_$v.log
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03aa_bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['kernel_breakdown_ms'].items() if 'attn' in k})
except Exception as e:
    print('$v', 'failed', e); print(open('gpurun_out/r03aa_bench_$v.log').read()[-800:])
PY
done
unset MBX_LIB
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03aa_pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/r03aa_pytest_gpu.log
