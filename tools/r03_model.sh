#!/bin/bash
export RTC_AMD_LIB=$PWD/ray_tracer_challenge_amd/librtc_amd_dev.so RTC_AMD_JIT_PRINT=1
for sc in "soft_shadows 4096" "glass_and_mirror 4096" "reflect_refract 4096x2048" "first_textures 4096x2048" "hexagons 4096x2048" "first_scene 4096x2048" "first_plane 4096x2048" "first_patterns 4096x2048" "skybox 4096x2048" "grouped_grid 4096" "soft_shadows 2048" "single_sphere 1024"; do set -- $sc
python tools/time_scene.py $1 $2 5 2>&1 | grep "modelled\|kernel_ms"; done
