#!/bin/bash
# the development switches (RTC_AMD_JIT_FLAGS, _BLOCK_S, ...) exist only in the development build of the library
export RTC_AMD_LIB="${RTC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ray_tracer_challenge_amd/librtc_amd_dev.so}"
# development: occupancy target of the scene-compiled kernel, interleaved on one box:  tools/ab_waves.sh soft_shadows 4096
for r in 1 2; do
  for w in 5 6 7 8; do RTC_AMD_JIT_FLAGS="-DRTC_WAVES_PER_SIMD=$w" python tools/time_scene.py "$@" 20 2>&1 | grep -v amdgpu.ids | sed "s/^/[waves $w] /"; done
done
