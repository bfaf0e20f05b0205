#!/bin/bash
# Round 3, session G: LayerNorm folding with the row dots taken at the staging step -- kernel tests, model tests, A/B, kernel trace.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fold.py tests/test_gpu_kernels.py -q -p no:cacheprovider -k "fold or attn or stats or lnbwd or identity or plain" > gpurun_out/r03g_pytest_fold.log 2>&1
echo "pytest fold exit $?" > gpurun_out/r03g_summary.txt
timeout 600 python -m pytest tests/test_gpu_model.py -q -p no:cacheprovider -k "tiny_golden or baseline_shape or full_size or recompute_mode" > gpurun_out/r03g_pytest_model.log 2>&1
echo "pytest model exit $?" >> gpurun_out/r03g_summary.txt
MBX_FOLD_LN=0 timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03g_bench_nofold.json 2> gpurun_out/r03g_bench_nofold.log
timeout 300 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r03g_bench_fold.json 2> gpurun_out/r03g_bench_fold.log
cd /tmp; rm -rf /tmp/kt
( cd $GRAFT_REPO_ROOT && MBX_DUAL_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r03g_prof.log 2>&1 )
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > gpurun_out/r03g_kernel_stats.txt 2>&1
cat gpurun_out/r03g_summary.txt; tail -6 gpurun_out/r03g_pytest_fold.log | cut -c1-400; tail -6 gpurun_out/r03g_pytest_model.log | cut -c1-400; head -16 gpurun_out/r03g_kernel_stats.txt | cut -c1-160
for v in nofold fold; do python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03g_bench_$v.json').read().strip().splitlines()[-1])
    kb = d.get('kernel_breakdown_ms', {})
    print('$v', d['value'], d['ms_per_step'], {k: (kb[k]['calls'], kb[k]['ms']) for k in list(kb)[:9]})
except Exception as e:
    print('$v', 'failed', e); print(open('gpurun_out/r03g_bench_$v.log').read()[-1500:])
PY
done
