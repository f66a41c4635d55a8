#!/bin/bash
# the development switches (RTC_AMD_JIT_FLAGS, _BLOCK_S, ...) exist only in the development build of the library
export RTC_AMD_LIB="${RTC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ray_tracer_challenge_amd/librtc_amd_dev.so}"
# development: occupancy target of the scene-compiled traversal kernels
for sc in "sphere_grid 8192" "hexagons 4096" "grouped_grid 4096" "mesh 2048" "here_be_dragons 2000x800"; do
  for w in 4 5 6 7 8; do RTC_AMD_TREE_WAVES=$w python tools/time_scene.py $sc 5 2>&1 | grep -v amdgpu.ids | sed "s/^/[tree waves $w] /"; done
done
