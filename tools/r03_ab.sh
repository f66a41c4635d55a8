#!/bin/bash
# same-box A/B of the frame scheduling switches on the scenes they matter for (development)
mkdir -p gpurun_out/r03z
for sc in "mesh 2048 2048" "here_be_dragons 4000 1600" "hexagons 4096 2048" "grouped_grid 4096 4096" "sphere_grid 8192 8192"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 10 --rounds 2 "default" "no feedback|RTC_AMD_BLOCK_FEEDBACK=0" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03z/ab_q.txt || exit 1
done
