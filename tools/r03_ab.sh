#!/bin/bash
mkdir -p gpurun_out/r03r
P5="RTC_AMD_JIT_FLAGS=-DRTC_SPEC_STASH=0 -DRTC_SPEC_LDS_FRAMES=5 -DRTC_WAVES_PER_SIMD=5"
P6="RTC_AMD_JIT_FLAGS=-DRTC_SPEC_STASH=0 -DRTC_SPEC_LDS_FRAMES=5 -DRTC_WAVES_PER_SIMD=6"
P5N="RTC_AMD_JIT_FLAGS=-DRTC_SPEC_STASH=0 -DRTC_WAVES_PER_SIMD=5"
for sc in "reflect_refract 4096 2048" "first_scene 4096 2048" "first_plane 4096 2048" "first_patterns 4096 2048" "skybox 4096 2048" "shapes_medley 2048 1536"; do set -- $sc
python tools/ab_env.py --scene $1 --size $2 --height $3 --steps 8 --rounds 2 "default" "nostash lds5 w5|$P5" "nostash lds5 w6|$P6" "nostash w5|$P5N" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r03r/ab.txt
done
