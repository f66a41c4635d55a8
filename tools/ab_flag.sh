#!/bin/bash
# development: A/B of one hiprtc flag over several scenes, interleaved on one box:  tools/ab_flag.sh "-DFOO" "soft_shadows 4096" "mesh 2048" ...
FLAGS="$1"; shift
for sc in "$@"; do
  for r in 1 2; do
    RTC_AMD_JIT_FLAGS="$FLAGS" python tools/time_scene.py $sc 10 2>&1 | grep -v amdgpu.ids | sed "s/^/[with $FLAGS] /"
    python tools/time_scene.py $sc 10 2>&1 | grep -v amdgpu.ids | sed "s/^/[default] /"
  done
done
