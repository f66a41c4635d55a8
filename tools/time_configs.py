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
def timed(world, camera, depth):
    r = Renderer(world, camera, device=0); out = r.alloc()
    for _ in range(10): r.render(depth, out=out)   # (ten warm-up launches: after a run of tiny kernels the clocks need them -- with two,
    r.stats()                                       #  C3 read 0.917 ms where bench.py on the same box measured 0.826)
    for _ in range(10): r.render(depth, out=out)
    st = r.stats()
    name = r.kernel_name
    r.close()
    return st, name


print("| scene | kernel | kernel ms (frames scheduled by the frame before) | kernel ms (every frame like the first: RTC_AMD_BLOCK_FEEDBACK=0) | rays per frame | Grays/s | Gpixel/s |")
print("|---|---|---|---|---|---|---|")
for label, name, size in CASES:
    world, camera, depth = getattr(scenes, name)(*size)
    os.environ.pop("RTC_AMD_BLOCK_FEEDBACK", None)
    st, kernel = timed(world, camera, depth)
    os.environ["RTC_AMD_BLOCK_FEEDBACK"] = "0"   # (switches are read when a context is created)
    st0, _ = timed(world, camera, depth)
    os.environ.pop("RTC_AMD_BLOCK_FEEDBACK", None)
    assert st0["rays"] == st["rays"]
    print("| %s | %s | %.3f | %.3f | %d | %.1f | %.2f |" % (label, kernel, st["kernel_ms"], st0["kernel_ms"], st["rays"], st["rays"] / st["kernel_ms"] / 1e6,
                                                         st["pixels"] / st["kernel_ms"] / 1e6), flush=True)
