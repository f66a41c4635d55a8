#!/bin/bash
mkdir -p gpurun_out/r03z
for sc in "mesh 2048 2048" "here_be_dragons 4000 1600" "mesh 3072 3072"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 5 --rounds 2 "default" "first frames|RTC_AMD_BLOCK_FEEDBACK=0" "first frames, two lanes on all mesh tiles|RTC_AMD_BLOCK_FEEDBACK=0|RTC_AMD_BLOCK_S=1" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03z/ab18.txt || exit 1
done
