#!/bin/bash
mkdir -p gpurun_out/r03s
P5="RTC_AMD_JIT_FLAGS=-DRTC_SPEC_STASH=0 -DRTC_SPEC_LDS_FRAMES=5|RTC_AMD_TREE_WAVES=5"
P6="RTC_AMD_JIT_FLAGS=-DRTC_SPEC_STASH=0 -DRTC_SPEC_LDS_FRAMES=5"
W5="RTC_AMD_TREE_WAVES=5"
for sc in "hexagons 4096 2048" "grouped_grid 4096 4096" "sphere_grid 8192 8192" "mesh 2048 2048" "here_be_dragons 4000 1600"; do set -- $sc
python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 5 --rounds 2 "default" "nostash lds5 w5|$P5" "nostash lds5 w6|$P6" "w5|$W5" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03s/ab.txt
done
