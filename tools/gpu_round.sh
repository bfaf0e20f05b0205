#!/bin/bash
# One GPU-box session: per-group pytest (separate processes so that a fault in one kernel does not
# hide the others), smoke, a short bench.  Everything lands in gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.txt
for grp in embed layernorm fuse head prep_weights gemm_nt_store gemm_nt_epilogues gemm_tn attention; do
  timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "$grp" -p no:cacheprovider > gpurun_out/k_$grp.log 2>&1
  echo "$grp exit $?" >> gpurun_out/summary.txt
  cp gpurun_out/kernel_parity.json gpurun_out/kernel_parity_$grp.json 2>/dev/null
done
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -p no:cacheprovider > gpurun_out/model.log 2>&1
echo "model exit $?" >> gpurun_out/summary.txt
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/summary.txt
timeout 600 python bench.py --steps 5 --warmup 2 ${BENCH_ARGS} > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/summary.txt
cat gpurun_out/summary.txt
tail -3 gpurun_out/bench.log
