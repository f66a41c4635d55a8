#!/usr/bin/env python3
"""Kernel time of every BASELINE configuration and demo scene (development / documentation tool)."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.renderer import Renderer
CASES = [("C1 soft_shadows 1000x400", "soft_shadows", (1000, 400)), ("C2 single_sphere 1024^2", "single_sphere", (1024, 1024)),
         ("C3 soft_shadows 4096^2", "soft_shadows", (4096, 4096)), ("C4 glass_and_mirror 4096^2", "glass_and_mirror", (4096, 4096)),
         ("C5 sphere_grid 8192^2", "sphere_grid", (8192, 8192)), ("first_scene 4096x2048", "first_scene", (4096, 2048)),
         ("first_plane 4096x2048", "first_plane", (4096, 2048)), ("first_patterns 4096x2048", "first_patterns", (4096, 2048)),
         ("reflect_refract 4096x2048", "reflect_refract", (4096, 2048)), ("hexagons 4096x2048", "hexagons", (4096, 2048)),
         ("first_textures 4096x2048", "first_textures", (4096, 2048)), ("skybox 4096x2048", "skybox", (4096, 2048)),
         ("grouped_grid 4096^2", "grouped_grid", (4096, 4096)), ("mesh 2048^2", "mesh", (2048, 2048)),
         ("mesh 512x384", "mesh", (512, 384)), ("here_be_dragons 1000x400 (17.9 k triangles)", "here_be_dragons", (1000, 400)),
         ("here_be_dragons 4000x1600", "here_be_dragons", (4000, 1600))]
def make(world, camera, feedback):
    if feedback: os.environ.pop("RTC_AMD_BLOCK_FEEDBACK", None)
    else: os.environ["RTC_AMD_BLOCK_FEEDBACK"] = "0"   # (switches are read when a context is created)
    r = Renderer(world, camera, device=0)
    os.environ.pop("RTC_AMD_BLOCK_FEEDBACK", None)
    return r


print("| scene | kernel | kernel ms (frames scheduled by the frame before) | kernel ms (every frame like the first: RTC_AMD_BLOCK_FEEDBACK=0) | rays per frame | Grays/s | Gpixel/s |")
print("|---|---|---|---|---|---|---|")
for label, name, size in CASES:
    world, camera, depth = getattr(scenes, name)(*size)
    # both contexts side by side, their measurements interleaved (one after the other, the second read 2 - 4 % faster on
    # scenes where both run the very same launches: clocks), ten warm-up launches each, the median of five rounds of ten
    rs = [make(world, camera, True), make(world, camera, False)]
    outs = [r.alloc() for r in rs]
    for r, out in zip(rs, outs):
        for _ in range(10): r.render(depth, out=out)
        r.stats()
    ms = [[], []]
    for _ in range(5):
        for k in (0, 1):
            for _ in range(10): rs[k].render(depth, out=outs[k])
            st = rs[k].stats()
            ms[k].append(st["kernel_ms"])
            last = st
    kernel = rs[0].kernel_name
    for r in rs: r.close()
    a, b = sorted(ms[0])[2], sorted(ms[1])[2]
    print("| %s | %s | %.3f | %.3f | %d | %.1f | %.2f |" % (label, kernel, a, b, last["rays"], last["rays"] / a / 1e6, last["pixels"] / a / 1e6), flush=True)
