#!/bin/bash
mkdir -p gpurun_out/r03z
for sc in "here_be_dragons 1000 400" "mesh 512 384" "mesh 1024 1024" "mesh 2048 2048" "here_be_dragons 4000 1600" "soft_shadows 1000 400" "soft_shadows 512 512" "patterns_medley 1024 1024"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 10 --rounds 2 "up to 16 lanes" "up to 8 lanes|RTC_AMD_FEEDBACK_MAX_S=3" "16 lanes, 3 passes|RTC_AMD_FEEDBACK_PASSES=3" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03z/ab15.txt || exit 1
done
