#!/bin/bash
mkdir -p gpurun_out/r03v
for sc in "mesh 512 384" "mesh 1024 1024" "mesh 2048 2048"; do set -- $sc
timeout -k 10 300 python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 5 --rounds 2 "default" "no split|RTC_AMD_TREE_SPLIT=0" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03v/ab.txt || exit 1
done
