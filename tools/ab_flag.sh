#!/bin/bash
# the development switches (RTC_AMD_JIT_FLAGS, _BLOCK_S, ...) exist only in the development build of the library
export RTC_AMD_LIB="${RTC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ray_tracer_challenge_amd/librtc_amd_dev.so}"
# development: A/B of one hiprtc flag over several scenes, interleaved on one box:  tools/ab_flag.sh "-DFOO" "soft_shadows 4096" "mesh 2048" ...
FLAGS="$1"; shift
for sc in "$@"; do
  for r in 1 2; do
    RTC_AMD_JIT_FLAGS="$FLAGS" python tools/time_scene.py $sc 10 2>&1 | grep -v amdgpu.ids | sed "s/^/[with $FLAGS] /"
    python tools/time_scene.py $sc 10 2>&1 | grep -v amdgpu.ids | sed "s/^/[default] /"
  done
done
