#!/bin/bash
# Round 3, session AO: does the Infinity Cache keep what the producer wrote last?  LayerNorm forward walking its rows from the END
# (-DMBX_LN_REV=1) against the product kernel, inside the real step (per-entry times of the bench line).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
for v in base lnrev base lnrev; do
  unset MBX_LIB; [ $v != base ] && export MBX_LIB=tools/variants/libmbx_$v.so
  timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03ao_$v.json 2> gpurun_out/r03ao_$v.log
  python - <<PY
import json
d = json.loads(open('gpurun_out/r03ao_$v.json').read().strip().splitlines()[-1]); kb = d['kernel_breakdown_ms']
print('$v', d['value'], d['ms_per_step'], 'ln_fwd', kb['layernorm_fwd']['ms'], 'gemm_nt', kb['gemm_nt']['ms'])
PY
done
