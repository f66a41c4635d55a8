#!/bin/bash
mkdir -p gpurun_out/r03z
for sc in "glass_and_mirror 4096 4096" "reflect_refract 4096 2048" "soft_shadows 4096 4096" "hexagons 4096 2048" "first_textures 4096 2048" "sphere_grid 8192 8192" "first_plane 4096 2048" "mesh 2048 2048"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 10 --rounds 3 "nt stores" "plain stores|RTC_AMD_JIT_SOURCE=tools/ab_core_old.h" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03z/ab13.txt || exit 1
done
