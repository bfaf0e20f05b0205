#!/bin/bash
# Round profiles on the GPU box: rocprofv3 kernel trace (+stats) and the three separate PMC passes of the SAME bench command,
# summarised into gpurun_out/${TAG}_kernel_stats.txt and gpurun_out/${TAG}_pmc_bench.txt (copy both into profiles/).
#   gpurun --timeout 1500 -- 'TAG=r02 bash tools/gpu_profiles.sh'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TAG=${TAG:-r02}
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp MBX_DUAL_STREAM=0      # one stream: per-kernel durations are not inflated by overlap
CMD="python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline"
cd /tmp
rm -rf /tmp/kt
( cd $GRAFT_REPO_ROOT && timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- $CMD > gpurun_out/${TAG}_prof.log 2>&1 ); echo "kernel-trace exit $?"
DB=$(find /tmp/kt -name "*.db" | head -1)
cd $GRAFT_REPO_ROOT
{ echo "# command: MBX_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- $CMD"; python tools/rocpd_stats.py $DB 40; } > gpurun_out/${TAG}_kernel_stats.txt
tail -1 gpurun_out/${TAG}_prof.log | cut -c1-300
PCMD="python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline"
run() { tag=$1; shift; rm -rf /tmp/pmcb; timeout 400 rocprofv3 --pmc "$@" -d /tmp/pmcb -o p -- $PCMD > /dev/null 2>&1; python tools/pmc_stats.py $(find /tmp/pmcb -name "*.db" | head -1) "" | grep -E "gemm|attn|ln_|fuse|adamw|pose|embed|head|colsum|prep|fold|rowc" > gpurun_out/pmc_bench_$tag.txt; echo "pmc $tag: $(wc -l < gpurun_out/pmc_bench_$tag.txt) rows"; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE TCC_HIT_sum
run write WRITE_SIZE TCC_MISS_sum
python tools/pmc_table.py gpurun_out gpurun_out/${TAG}_kernel_stats.txt > gpurun_out/${TAG}_pmc_bench.txt
head -30 gpurun_out/${TAG}_kernel_stats.txt; cat gpurun_out/${TAG}_pmc_bench.txt
