#!/bin/bash
mkdir -p gpurun_out/r03i
python tools/ab_env.py --scene soft_shadows --size 4096 --steps 20 --rounds 3 "signs" "sqrt_form|RTC_AMD_JIT_FLAGS=-DRTC_FAST_SQRT_FORM" "signs w6|RTC_AMD_JIT_FLAGS=-DRTC_WAVES_PER_SIMD=6" "signs w8|RTC_AMD_JIT_FLAGS=-DRTC_WAVES_PER_SIMD=8" "no_fast|RTC_AMD_FAST_SHADOW=0" > gpurun_out/r03i/ab.txt 2>&1; cat gpurun_out/r03i/ab.txt
for f in "" "-DRTC_FAST_SQRT_FORM"; do
RTC_AMD_LIB=ray_tracer_challenge_amd/librtc_amd_dev.so RTC_AMD_JIT_FLAGS="-DRTC_COUNT_EXACT $f" python - <<'PY' 2>&1 | grep -v amdgpu
import sys, os; sys.path.insert(0,'.')
from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.renderer import Renderer
for name,size in (("soft_shadows",(4096,4096)),("soft_shadows",(1000,400)),("first_textures",(2048,1024))):
    w,c,d=getattr(scenes,name)(*size)
    r=Renderer(w,c,device=0); r.render(d); st=r.stats(); print(os.environ["RTC_AMD_JIT_FLAGS"], name, size, "exact-path samples:", st["culled_shadow_rays"], "of rays", st["rays"])
PY
done
