#!/bin/bash
# Round 3, session E: LayerNorm folding -- new kernel tests, model fixtures, step A/B (MBX_FOLD_LN=0 / 1).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fold.py -q -p no:cacheprovider -x > gpurun_out/r03e_pytest_fold.log 2>&1
echo "pytest fold exit $?" > gpurun_out/r03e_summary.txt
timeout 600 python -m pytest tests/test_gpu_model.py -q -p no:cacheprovider -k "tiny_golden or baseline_shape or full_size or recompute_mode" > gpurun_out/r03e_pytest_model.log 2>&1
echo "pytest model exit $?" >> gpurun_out/r03e_summary.txt
MBX_FOLD_LN=0 timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03e_bench_nofold.json 2> gpurun_out/r03e_bench_nofold.log
timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03e_bench_fold.json 2> gpurun_out/r03e_bench_fold.log
cat gpurun_out/r03e_summary.txt; tail -25 gpurun_out/r03e_pytest_fold.log; tail -8 gpurun_out/r03e_pytest_model.log
for v in nofold fold; do python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03e_bench_$v.json').read().strip().splitlines()[-1])
    kb = d.get('kernel_breakdown_ms', {})
    print('$v', d['value'], d['ms_per_step'], {k: (kb[k]['calls'], kb[k]['ms']) for k in kb})
except Exception as e:
    print('$v', 'failed', e); print(open('gpurun_out/r03e_bench_$v.log').read()[-1500:])
PY
done
