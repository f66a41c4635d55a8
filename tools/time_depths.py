#!/usr/bin/env python3
"""Kernel time of one scene at recursion depths 0..5 (development: where does a frame's time go?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.renderer import Renderer
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
world, camera, _ = getattr(scenes, name)(w, h)
r = Renderer(world, camera, device=0)
out = r.alloc()
for depth in range(6):
    for _ in range(2): r.render(depth, out=out)
    r.stats()
    for _ in range(4): r.render(depth, out=out)
    st = r.stats()
    print("%s %dx%d depth %d: %.3f ms, %d rays, %d shaded hits" % (name, w, h, depth, st["kernel_ms"], st["rays"], st["shaded_hits"]), flush=True)
