#!/bin/bash
# Round 3, session K: (1) the north-star Block configuration (B=256) under rocprofv3: kernel trace + the three PMC passes;
# (2) HBM-side traffic of the weight-gradient GEMM per shape (FETCH_SIZE / WRITE_SIZE / L2 hit).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp MBX_DUAL_STREAM=0
CMD="python bench.py --block"
cd /tmp; rm -rf /tmp/ktb
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktb -o kt -- $CMD > gpurun_out/r03k_block.log 2>&1 ); echo "block trace exit $?"
cd $GRAFT_REPO_ROOT
{ echo "# command: MBX_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- $CMD   (one Block, B=256 T=243 J=17 C=512, bf16: 1 + 5 forward passes, 1 + 5 forward+backward passes)"; python tools/rocpd_stats.py $(find /tmp/ktb -name "*.db" | head -1) 40; } > gpurun_out/r03k_block_kernel_stats.txt
run() { tag=$1; shift; rm -rf /tmp/pmcb; timeout 300 rocprofv3 --pmc "$@" -d /tmp/pmcb -o p -- $CMD > /dev/null 2>&1; python tools/pmc_stats.py $(find /tmp/pmcb -name "*.db" | head -1) "" | grep -E "gemm|attn|ln_|fuse|colsum|fold|rowc" > gpurun_out/pmc_bench_$tag.txt; echo "pmc $tag: $(wc -l < gpurun_out/pmc_bench_$tag.txt) rows"; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE TCC_HIT_sum
run write WRITE_SIZE TCC_MISS_sum
python tools/pmc_table.py gpurun_out gpurun_out/r03k_block_kernel_stats.txt | sed 's/one bench step + warm-up + instrumented step, MBX_DUAL_STREAM=0, 64 clips x 243 frames, bf16/bench.py --block: one Block at B=256 (forward x6, forward+backward x6), MBX_DUAL_STREAM=0, bf16/' > gpurun_out/r03k_block_pmc.txt
tail -2 gpurun_out/r03k_block.log | cut -c1-700; head -24 gpurun_out/r03k_block_kernel_stats.txt | cut -c1-150; cat gpurun_out/r03k_block_pmc.txt | cut -c1-170
# (2) weight-gradient GEMM per shape
unset MBX_DUAL_STREAM
for shp in dW_qkv dW_proj dW_fc1 dW_fc2; do
  for pass in fetch write; do
    rm -rf /tmp/pmcg
    if [ $pass = fetch ]; then C="FETCH_SIZE TCC_HIT_sum"; else C="WRITE_SIZE TCC_MISS_sum"; fi
    timeout 200 rocprofv3 --pmc $C -d /tmp/pmcg -o p -- python tools/gemm_bench.py --iters 2 --check 0 --only $shp > /dev/null 2>&1
    echo "== $shp $pass"; python tools/pmc_stats.py $(find /tmp/pmcg -name "*.db" | head -1) gemm_tn
  done
done > gpurun_out/r03k_tn_traffic.txt 2>&1
cat gpurun_out/r03k_tn_traffic.txt | cut -c1-140
