#!/bin/bash
mkdir -p gpurun_out/r03z
for sc in "soft_shadows 2048 2048" "soft_shadows 3072 3072" "soft_shadows 4096 4096" "first_textures 4096 2048"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 10 --rounds 2 "default" "lists up to 300k waves|RTC_AMD_AREA_SHARE_WAVES=300000" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03z/ab11.txt || exit 1
done
