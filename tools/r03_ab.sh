#!/bin/bash
mkdir -p gpurun_out/r03y
for sc in "mesh 512 384" "mesh 1024 1024" "mesh 2048 2048" "here_be_dragons 1000 400" "here_be_dragons 4000 1600"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 5 --rounds 2 "1 pass|RTC_AMD_FEEDBACK_PASSES=1" "2 passes" "3 passes|RTC_AMD_FEEDBACK_PASSES=3" "no feedback|RTC_AMD_BLOCK_FEEDBACK=0" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03y/ab4.txt || exit 1
done
