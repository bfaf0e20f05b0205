#!/bin/bash
# Round 3, session AK: the fused residual GEMM + LayerNorm at small batches (where launches and L2 residency weigh differently).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
for b in 4 16; do for f in 0 1; do
  MBX_RESID_LN=$f timeout 200 python bench.py --batch $b --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r03ak_b${b}_f$f.json 2> gpurun_out/r03ak_b${b}_f$f.log
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03ak_b${b}_f$f.json').read().strip().splitlines()[-1]); print('batch $b resid_ln $f', d['value'], d['ms_per_step'])
except Exception as e:
    print('batch $b resid_ln $f failed', e); print(open('gpurun_out/r03ak_b${b}_f$f.log').read()[-500:])
PY
done; done
