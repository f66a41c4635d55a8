#!/usr/bin/env python3
"""kernel time of one scene/size (development tool): python tools/time_scene.py sphere_grid 8192|1000x400 [steps] [key=value ...]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.renderer import Renderer
name, size = sys.argv[1], sys.argv[2]; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
kw = {k: int(v) for k, v in (a.split("=") for a in sys.argv[4:])}
w, h = (int(v) for v in size.split("x")) if "x" in size else (int(size), int(size))
world, camera, depth = getattr(scenes, name)(w, h, **kw)
r = Renderer(world, camera, device=0); out = r.alloc()
for _ in range(2): r.render(depth, out=out)
r.stats()
for _ in range(steps): r.render(depth, out=out)
st = r.stats()
print(name, size, kw, r.kernel_name, "kernel_ms %.4f" % st["kernel_ms"], "rays", st["rays"], "Grays/s %.1f" % (st["rays"] / st["kernel_ms"] / 1e6),
      "hash", hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12])
