#!/bin/bash
mkdir -p gpurun_out/r03z
for sc in "soft_shadows 4096 4096" "glass_and_mirror 4096 4096" "reflect_refract 4096 2048" "first_scene 4096 2048" "hexagons 4096 2048" "mesh 2048 2048"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 10 --rounds 2 "default" "interleave 10|RTC_AMD_FEEDBACK_INTERLEAVE_PCT=10" "interleave 25|RTC_AMD_FEEDBACK_INTERLEAVE_PCT=25" "interleave 50|RTC_AMD_FEEDBACK_INTERLEAVE_PCT=50" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03z/ab5.txt || exit 1
done
