#!/bin/bash
mkdir -p gpurun_out/r03u
for sc in "mesh 2048 2048" "hexagons 4096 2048" "here_be_dragons 4000 1600"; do set -- $sc
python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 5 --rounds 2 "default" "second_walk|RTC_AMD_JIT_FLAGS=-DRTC_NO_MERGED_N12" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03u/ab.txt
done
