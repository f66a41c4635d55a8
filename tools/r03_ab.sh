#!/bin/bash
mkdir -p gpurun_out/r03z
for sc in "soft_shadows 4096 4096" "glass_and_mirror 4096 4096" "reflect_refract 4096 2048" "first_textures 4096 2048" "hexagons 4096 2048"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 10 --rounds 2 "mode 1|RTC_AMD_GRID_FEEDBACK=0" "mode 2|RTC_AMD_GRID_FEEDBACK=0|RTC_AMD_SWIZZLE_MODE=2" "mode 3|RTC_AMD_GRID_FEEDBACK=0|RTC_AMD_SWIZZLE_MODE=3" "mode 4|RTC_AMD_GRID_FEEDBACK=0|RTC_AMD_SWIZZLE_MODE=4" "none|RTC_AMD_GRID_FEEDBACK=0|RTC_AMD_SWIZZLE=0" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03z/ab8.txt || exit 1
done
