#!/bin/bash
# PMC passes over the GEMM micro-benchmark (separate passes: TCC slots are limited; never combined with tracing)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
ONLY=${ONLY:-qkv,proj,dW_qkv}
PASSES=${PASSES:-"fetch write sq1 sq2"}
run() { tag=$1; shift; timeout 300 rocprofv3 --pmc "$@" -d gpurun_out/pmc_$tag -o pmc -- python tools/gemm_bench.py --iters 2 --check 0 --only $ONLY > gpurun_out/pmc_$tag.log 2>&1; echo "pmc $tag exit $?"; }
for t in $PASSES; do
  case $t in
    fetch) run fetch FETCH_SIZE TCC_HIT_sum;;
    write) run write WRITE_SIZE TCC_MISS_sum;;
    sq1) run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE;;
    sq2) run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS;;
  esac
  f=$(find gpurun_out/pmc_$t -name "*.db" | head -1); echo "== $t"; python tools/pmc_stats.py $f gemm | grep -v colsum
done
