#!/bin/bash
# the development switches (RTC_AMD_JIT_FLAGS, _BLOCK_S, ...) exist only in the development build of the library
export RTC_AMD_LIB="${RTC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ray_tracer_challenge_amd/librtc_amd_dev.so}"
# development: lanes per pixel (RTC_AMD_SHARE_LOG2) on area-light scenes at several sizes
for sc in "soft_shadows 1000x400" "soft_shadows 512" "soft_shadows 2048" "soft_shadows 4096" "first_textures 1000x500"; do
  for s in 0 1 2 3; do RTC_AMD_SHARE_LOG2=$s python tools/time_scene.py $sc 10 2>&1 | grep -v amdgpu.ids | sed "s/^/[share $s] /" || exit 1; done
done
