#!/bin/bash
# same-box A/B of the frame scheduling switches on the scenes they matter for (development)
mkdir -p gpurun_out/r03z
for sc in "soft_shadows 1000 400" "soft_shadows 4096 4096" "reflect_refract 4096 2048" "first_textures 4096 2048" "mesh 2048 2048" "here_be_dragons 4000 1600"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 10 --rounds 2 "default" "no feedback|RTC_AMD_BLOCK_FEEDBACK=0" "no feedback, image order|RTC_AMD_BLOCK_FEEDBACK=0|RTC_AMD_SWIZZLE=0" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03z/ab_final.txt || exit 1
done
