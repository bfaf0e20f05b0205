#!/bin/bash
# Round 3, session I: batched LDS reads in the staging dots (kernel tests + trace), wide residual epilogue A/B, CU-mask probe.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fold.py -q -p no:cacheprovider > gpurun_out/r03i_pytest_fold.log 2>&1
echo "pytest fold exit $?" > gpurun_out/r03i_summary.txt
timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03i_bench_fold.json 2> gpurun_out/r03i_bench_fold.log
MBX_LIB=tools/variants/libmbx_resid_wide.so timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03i_bench_wide.json 2> gpurun_out/r03i_bench_wide.log
MBX_LIB=tools/variants/libmbx_resid_wide.so timeout 300 python tools/gemm_bench.py --only proj,fc2 > gpurun_out/r03i_gemm_wide.txt 2>&1
timeout 300 python tools/gemm_bench.py --only proj,fc2 > gpurun_out/r03i_gemm_base.txt 2>&1
timeout 400 python tools/cu_mask_probe.py > gpurun_out/r03i_cu_mask.txt 2>&1
cd /tmp; rm -rf /tmp/kt
( cd $GRAFT_REPO_ROOT && MBX_DUAL_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r03i_prof.log 2>&1 )
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) > gpurun_out/r03i_kernel_stats.txt 2>&1
cat gpurun_out/r03i_summary.txt; tail -3 gpurun_out/r03i_pytest_fold.log | cut -c1-300
grep "^nt" gpurun_out/r03i_gemm_base.txt gpurun_out/r03i_gemm_wide.txt; cat gpurun_out/r03i_cu_mask.txt | tail -8; grep "attn_bwd" gpurun_out/r03i_kernel_stats.txt | cut -c1-120
for v in fold wide; do python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03i_bench_$v.json').read().strip().splitlines()[-1])
    kb = d.get('kernel_breakdown_ms', {})
    print('$v', d['value'], d['ms_per_step'], {k: (kb[k]['calls'], kb[k]['ms']) for k in list(kb)[:6]})
except Exception as e:
    print('$v', 'failed', e); print(open('gpurun_out/r03i_bench_$v.log').read()[-1500:])
PY
done
