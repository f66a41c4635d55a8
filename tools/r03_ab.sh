#!/bin/bash
mkdir -p gpurun_out/r03z
for sc in "soft_shadows 1000 400" "soft_shadows 512 512" "soft_shadows 1024 1024" "soft_shadows 1536 1536" "soft_shadows 2048 2048" "patterns_medley 1024 1024"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 10 --rounds 2 "default" "no feedback|RTC_AMD_BLOCK_FEEDBACK=0" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03z/ab10.txt || exit 1
done
