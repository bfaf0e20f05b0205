#!/bin/bash
# compile ONE kernel source for gfx950 and print the compiler's per-kernel resource usage (registers, scratch, spills)
#   tools/cc1.sh mlp_fused.hip [-S]     (-S: also write /tmp/<name>.s)
src=/root/repo/motionbert_amd/csrc/$1
name=$(basename "$1" .hip)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c "$src" -o /tmp/$name.o 2>&1 | grep -v "^$" | grep -E "error|warning: |Function Name|VGPRs|AGPRs|Scratch|Spill|Occupancy" | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//' -e 's/^.*remark: //'
if [ "$2" = "-S" ]; then /opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only "$src" -o /tmp/$name.s 2>/dev/null; fi
