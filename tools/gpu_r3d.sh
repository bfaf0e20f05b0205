#!/bin/bash
# Round 3, session D: weight-gradient GEMM A/B -- v0 = shipped round-2 loop (builtin transpose reads: the compiler drains the
# LDS-DMA ring with vmcnt(0) every chunk), v1 = same ping-pong loop with asm reads, v2 (product) = lock-step half-chunk pipeline.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
TN=dW_qkv,dW_proj,dW_fc1,dW_fc2
for v in tn_v0 tn_v1; do
  MBX_LIB=tools/variants/libmbx_$v.so timeout 300 python tools/gemm_bench.py --only $TN > gpurun_out/r03d_gemm_$v.txt 2>&1
done
timeout 300 python tools/gemm_bench.py --only $TN > gpurun_out/r03d_gemm_tn_v2.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -k "tn or gemm" > gpurun_out/r03d_pytest_gemm.log 2>&1
echo "pytest gemm exit $?" > gpurun_out/r03d_summary.txt
MBX_LIB=tools/variants/libmbx_tn_v0.so timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03d_bench_v0.json 2> gpurun_out/r03d_bench_v0.log
MBX_LIB=tools/variants/libmbx_tn_v1.so timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03d_bench_v1.json 2> gpurun_out/r03d_bench_v1.log
timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03d_bench_v2.json 2> gpurun_out/r03d_bench_v2.log
cat gpurun_out/r03d_summary.txt; tail -3 gpurun_out/r03d_pytest_gemm.log
for v in tn_v0 tn_v1 tn_v2; do echo "== $v"; grep "^tn" gpurun_out/r03d_gemm_$v.txt; done
for v in v0 v1 v2; do python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03d_bench_$v.json').read().strip().splitlines()[-1])
    kb = d.get('kernel_breakdown_ms', {})
    print('$v', d['value'], d['ms_per_step'], {k: kb[k]['ms'] for k in ('gemm_nt', 'gemm_tn') if k in kb})
except Exception as e:
    print('$v', 'failed', e)
PY
done
