#!/bin/bash
# round 3, development: does the light in VGPRs pay outside the headline kernel?
mkdir -p gpurun_out/r03f
LV="-DRTC_LIGHT_VGPRS"
python tools/ab_env.py --scene soft_shadows --size 1000 --height 400 --steps 30 --rounds 3 "default" "lv|RTC_AMD_JIT_FLAGS=$LV" > gpurun_out/r03f/ab_c1.txt 2>&1; cat gpurun_out/r03f/ab_c1.txt
python tools/ab_env.py --scene soft_shadows --size 2048 --steps 30 --rounds 3 "default" "lv|RTC_AMD_JIT_FLAGS=$LV" > gpurun_out/r03f/ab_2048.txt 2>&1; cat gpurun_out/r03f/ab_2048.txt
python tools/ab_env.py --scene first_textures --size 4096 --height 2048 --steps 10 --rounds 3 "default" "lv|RTC_AMD_JIT_FLAGS=$LV" "z0|RTC_AMD_JIT_FLAGS=-DRTC_SPEC_LIGHT_ZEROS=0" > gpurun_out/r03f/ab_ft.txt 2>&1; cat gpurun_out/r03f/ab_ft.txt
python tools/ab_env.py --scene patterns_medley --size 2048 --height 1536 --steps 10 --rounds 2 "default" "lv|RTC_AMD_JIT_FLAGS=$LV" > gpurun_out/r03f/ab_pm.txt 2>&1; cat gpurun_out/r03f/ab_pm.txt
