#!/usr/bin/env python3
"""development: the options a scene's kernel is compiled with, one line, as tools/spec_asm.sh takes them
    python tools/print_jit_flags.py mesh 2048 2048"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RTC_AMD_LIB", os.path.join(ROOT, "ray_tracer_challenge_amd", "librtc_amd_dev.so"))
os.environ["RTC_AMD_JIT_PRINT"] = "1"
sys.path.insert(0, ROOT)
from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.renderer import Renderer
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
world, camera, depth = getattr(scenes, name)(w, h)
r = Renderer(world, camera, device=0)
r.render(depth)
print(name, w, h, r.kernel_name, r.stats()["kernel_ms"])
