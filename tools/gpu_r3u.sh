#!/bin/bash
# Round 3, session U: residual GEMM + next LayerNorm, software-pipelined tail: parity, timing, whole step.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fold.py -x -q -k "xcd or resid_ln or fused_layernorm" > gpurun_out/r03u_pytest_rln.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r03u_pytest_rln.log
timeout 200 python tools/gemm_bench.py --only proj,fc2,proj_ln,fc2_ln,ln_only > gpurun_out/r03u_gemm.txt 2>&1; grep "^nt\|^  \|rror" gpurun_out/r03u_gemm.txt | cut -c1-110
for v in fused unfused; do
  export MBX_RESID_LN=1; [ $v = unfused ] && export MBX_RESID_LN=0
  timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03u_bench_$v.json 2> gpurun_out/r03u_bench_$v.log
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03u_bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])
except Exception as e:
    print('$v', 'failed', e); print(open('gpurun_out/r03u_bench_$v.log').read()[-800:])
PY
done
