#!/bin/bash
# Round 3, session N: operand-traffic knobs of the weight-gradient GEMM (chunks in flight, tile order, nt loads), dX-first order, wgrad stream.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
TN=dW_qkv,dW_proj,dW_fc1,dW_fc2
timeout 200 python tools/gemm_bench.py --only $TN > gpurun_out/r03n_gemm_base.txt 2>&1
for v in tn_a2 tn_o1 tn_nt tn_a2o1; do
  MBX_LIB=tools/variants/libmbx_$v.so timeout 200 python tools/gemm_bench.py --only $TN > gpurun_out/r03n_gemm_$v.txt 2>&1
done
timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03n_bench_base.json 2> gpurun_out/r03n_bench_base.log
MBX_FOLD_ORDER=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03n_bench_dxfirst.json 2> gpurun_out/r03n_bench_dxfirst.log
MBX_WGRAD_STREAM=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03n_bench_wgrad.json 2> gpurun_out/r03n_bench_wgrad.log
for v in base tn_a2 tn_o1 tn_nt tn_a2o1; do echo "== $v"; grep "^tn" gpurun_out/r03n_gemm_$v.txt; done
for v in base dxfirst wgrad; do python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03n_bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])
except Exception as e:
    print('$v', 'failed', e); print(open('gpurun_out/r03n_bench_$v.log').read()[-800:])
PY
done
