#!/bin/bash
# ONE parametrised GPU-box session (replaces the 40 one-shot tools/gpu_r3*.sh scripts of round 3 and the r4 ones):
#   gpurun --timeout 1800 -- 'bash tools/gpu_session.sh TAG step [step ...]'
# Every step writes gpurun_out/TAG_<step>.txt (merged back by gpurun) and prints its tail.  Steps:
#   pytest[:EXPR]        python -m pytest tests -m gpu [-k EXPR]            (EXPR: pytest -k expression, '+' for spaces)
#   file:PATH[:EXPR]     python -m pytest PATH [-k EXPR]
#   bench[:ARGS]         python bench.py ARGS                                 (ARGS: '+' for spaces; default: --no-cpu-baseline)
#   block | fwd[:CLIPS]  the north-star Block benchmark / forward-only throughput (tools/fwd_bench.py)
#   mlp[:CLIPS[:C]]      tools/mlp_time.py for the product library and every library in $VARIANTS (names under tools/variants/)
#   mlpbench | gemm      tools/mlp_bench.py / tools/gemm_bench.py
#   env:NAME=VALUE       export NAME=VALUE for the steps that follow (e.g. env:MBX_RAWLN=0, env:MBX_LIB=tools/variants/libmbx_x.so)
#   profiles[:CMD]       rocprofv3 kernel trace + the three separate PMC passes of CMD (default: the bench line's command, one
#                        stream) -> TAG_kernel_stats.txt, TAG_pmc.txt  (CMD: '+' for spaces; 'block' = bench.py --block)
#   py:SCRIPT[:ARGS]     python SCRIPT ARGS
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
TAG=$1; shift
sp() { echo "${1//+/ }"; }
out() { tee "gpurun_out/${TAG}_$1.txt" | grep -v "amdgpu.ids" | tail -${2:-12} | cut -c1-${3:-400}; }
n=0
for step in "$@"; do
  n=$((n + 1)); kind=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  echo "== [$TAG] $step"
  case $kind in
    env) export "$(sp "$arg")" ;;
    pytest) timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider ${arg:+-k "$(sp "$arg")"} 2>&1 | out pytest$n 8 300 ;;
    file) f=${arg%%:*}; k=""; [[ "$arg" == *:* ]] && k=${arg#*:}; timeout 1700 python -m pytest "$f" -q -p no:cacheprovider -x ${k:+-k "$(sp "$k")"} 2>&1 | out file$n 8 300 ;;
    bench) timeout 900 python bench.py $(sp "${arg:---no-cpu-baseline}") 2>gpurun_out/${TAG}_bench$n.log | out bench$n 2 4000 ;;
    block) timeout 600 python bench.py --block 2>/dev/null | out block$n 1 600 ;;
    fwd) timeout 600 python tools/fwd_bench.py $arg 2>&1 | out fwd$n 2 ;;
    mlp) a=(${arg//:/ }); { python tools/mlp_time.py ${a[0]:-256} ${a[1]:-512}; for v in $VARIANTS; do MBX_LIB=tools/variants/libmbx_$v.so python tools/mlp_time.py ${a[0]:-256} ${a[1]:-512}; done; } 2>&1 | out mlp$n 40 ;;
    mlpbench) timeout 600 python tools/mlp_bench.py $(sp "$arg") 2>&1 | out mlpbench$n 12 600 ;;
    gemm) timeout 600 python tools/gemm_bench.py $(sp "$arg") 2>&1 | out gemm$n 60 200 ;;
    py) s=${arg%%:*}; a=""; [[ "$arg" == *:* ]] && a=${arg#*:}; timeout 1200 python "$s" $(sp "$a") 2>&1 | out py$n 40 300 ;;
    profiles)
      CMD="python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline"; PCMD="python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline"
      [ "$arg" == "block" ] && CMD="python bench.py --block" && PCMD="$CMD"
      [ -n "$arg" ] && [ "$arg" != "block" ] && CMD="$(sp "$arg")" && PCMD="$CMD"
      export MBX_DUAL_STREAM=0      # one stream: per-kernel durations are not inflated by overlap
      ( cd /tmp; rm -rf /tmp/kt; cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- $CMD > gpurun_out/${TAG}_prof.log 2>&1 )
      DB=$(find /tmp/kt -name "*.db" | head -1)
      { echo "# command: MBX_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- $CMD"; python tools/rocpd_stats.py $DB 40; } > gpurun_out/${TAG}_kernel_stats.txt
      run() { t=$1; shift; rm -rf /tmp/pmcb; ( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --pmc "$@" -d /tmp/pmcb -o p -- $PCMD > /dev/null 2>&1 ); python tools/pmc_stats.py $(find /tmp/pmcb -name "*.db" | head -1) "" | grep -E "gemm|attn|ln_|fuse|mlp|rows_|adamw|pose|embed|head|colsum|prep|fold|rowc" > gpurun_out/pmc_bench_$t.txt; echo "pmc $t: $(wc -l < gpurun_out/pmc_bench_$t.txt) rows"; }
      run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
      run fetch FETCH_SIZE TCC_HIT_sum
      run write WRITE_SIZE TCC_MISS_sum
      { echo "# command (three separate --pmc passes): MBX_DUAL_STREAM=0 $PCMD"; python tools/pmc_table.py gpurun_out gpurun_out/${TAG}_kernel_stats.txt; } > gpurun_out/${TAG}_pmc.txt
      unset MBX_DUAL_STREAM
      head -24 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-170; cut -c1-170 gpurun_out/${TAG}_pmc.txt | head -30 ;;
    *) echo "unknown step $step" ;;
  esac
done
