#!/bin/bash
mkdir -p gpurun_out/r03o
timeout -k 10 400 python -m pytest tests/test_wavefront.py -m gpu -x -q 2>&1 | tail -4
for sc in "mesh 2048 2048" "here_be_dragons 4000 1600" "here_be_dragons 1000 400" "mesh 512 384" "hexagons 4096 2048" "mesh 1024 1024"; do set -- $sc
python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 5 --rounds 2 "per_pixel|RTC_AMD_WAVEFRONT=0" "by_levels|RTC_AMD_WAVEFRONT=1" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03o/ab.txt
done
