#!/bin/bash
# the development switches (RTC_AMD_JIT_FLAGS, _BLOCK_S, ...) exist only in the development build of the library
export RTC_AMD_LIB="${RTC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ray_tracer_challenge_amd/librtc_amd_dev.so}"
# development: the tree-kernel scenes
for sc in "sphere_grid 8192" "hexagons 4096" "grouped_grid 4096" "mesh 2048" "mesh 512" "first_textures 4096" "here_be_dragons 1000x400" "here_be_dragons 4000x1600"; do
  python tools/time_scene.py $sc 3 2>&1 | grep -v amdgpu.ids || exit 1
done
