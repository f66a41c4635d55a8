#!/usr/bin/env python3
"""development: kernel time of each 64-row band of a scene rendered alone, against the whole frame's --
shows whether a frame's time is the sum of its bands (throughput-bound) or close to its slowest band (a tail of
long-running waves).   python tools/band_times.py mesh 2048"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.renderer import Renderer
name, size = sys.argv[1], int(sys.argv[2])
world, camera, depth = getattr(scenes, name)(size, size)
r = Renderer(world, camera, device=0)
out = r.alloc()
for _ in range(3): r.render(depth, out=out)
r.stats()
r.render(depth, out=out)
whole = r.stats()["kernel_ms"]
n = (size + 63) // 64
ts = []
for b in range(n):
    p = Renderer.partition(64, n, b)
    o = r.alloc(p)
    r.render(depth, out=o, part=p); r.stats()
    r.render(depth, out=o, part=p)
    ts.append(r.stats()["kernel_ms"])
print(name, size, r.kernel_name, "whole %.3f ms; bands: sum %.3f max %.3f" % (whole, sum(ts), max(ts)))
print(" ".join("%.2f" % t for t in ts))
