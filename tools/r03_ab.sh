#!/bin/bash
mkdir -p gpurun_out/r03z
for sc in "soft_shadows 4096 4096" "glass_and_mirror 4096 4096" "first_scene 4096 2048"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 10 --rounds 2 "raster|RTC_AMD_GRID_FEEDBACK=0|RTC_AMD_SWIZZLE=0" "blocks_y 2|RTC_AMD_GRID_FEEDBACK=0|RTC_AMD_BLOCKS_Y=2" "blocks_y 4|RTC_AMD_GRID_FEEDBACK=0|RTC_AMD_BLOCKS_Y=4" "blocks_y 1 compiled in|RTC_AMD_GRID_FEEDBACK=0|RTC_AMD_BLOCKS_Y=1" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03z/ab9.txt || exit 1
done
