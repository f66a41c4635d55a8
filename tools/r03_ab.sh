#!/bin/bash
mkdir -p gpurun_out/r03z
for sc in "soft_shadows 4096 4096" "glass_and_mirror 4096 4096" "first_scene 4096 2048" "sphere_grid 8192 8192"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 10 --rounds 3 "default" "no grid feedback|RTC_AMD_GRID_FEEDBACK=0" "kernel without lists|RTC_AMD_GRID_FEEDBACK=0|RTC_AMD_JIT_SOURCE=tools/ab_core_notiles.h" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03z/ab4.txt || exit 1
done
