#!/bin/bash
# PMC passes over one short bench run (separate passes; never combined with tracing) -> gpurun_out/pmc_bench_*.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp MBX_DUAL_STREAM=0
run() { tag=$1; shift; rm -rf /tmp/pmcb; timeout 300 rocprofv3 --pmc "$@" -d /tmp/pmcb -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; python tools/pmc_stats.py $(find /tmp/pmcb -name "*.db" | head -1) "" | grep -E "gemm|attn|ln_|fuse" > gpurun_out/pmc_bench_$tag.txt; echo "pmc $tag: $(wc -l < gpurun_out/pmc_bench_$tag.txt) rows"; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE TCC_HIT_sum
run write WRITE_SIZE TCC_MISS_sum
