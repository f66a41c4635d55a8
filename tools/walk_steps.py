#!/usr/bin/env python3
"""What do the tree walks of a frame do, per pixel?  (development tool)

    python tools/walk_steps.py [--scene here_be_dragons --size 1000 --height 400] [ENV=value ...]

Renders once with -DRTC_DEBUG_STEPS: every pixel then holds its lane's counts over all the walks of the pixel -- group
boxes tested, leaf boxes tested (tri_precull), exact intersection tests, and the entries the lane's WAVE stepped through
(the union of its 64 lanes' paths).  One lane per pixel (RTC_AMD_SHARE_LOG2=0) unless the caller says otherwise.
"""
import os

# the development switches this tool drives exist only in the development build of the library
os.environ.setdefault("RTC_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ray_tracer_challenge_amd", "librtc_amd_dev.so"))
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(args):
    scene, w, h = "here_be_dragons", 1000, 0
    os.environ.setdefault("RTC_AMD_SHARE_LOG2", "0")
    it = iter(args)
    for a in it:
        if a == "--scene": scene = next(it)
        elif a == "--size": w = int(next(it))
        elif a == "--height": h = int(next(it))
        else:
            k, v = a.split("=", 1)
            os.environ[k] = v
    h = h or w
    os.environ["RTC_AMD_JIT_FLAGS"] = (os.environ.get("RTC_AMD_JIT_FLAGS", "") + " -DRTC_DEBUG_STEPS").strip()
    os.environ["RTC_AMD_SPECIALIZE"] = "1"
    from ray_tracer_challenge_amd import scenes
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = getattr(scenes, scene)(w, h)
    r = Renderer(world, camera, device=0)
    out = r.alloc()
    r.render(depth, out=out)
    st = r.stats()
    px = out.cpu().numpy().view(np.uint32).reshape(h, w, 3)[:h - 1, :w - 1]
    groups, exact, leaves, wave = px[..., 0] & 0xfffff, px[..., 0] >> 20, px[..., 1], px[..., 2]
    n = groups.size
    busy = wave > 0
    print("%s %dx%d  kernel %s  rays %d (%.2f per pixel)" % (scene, w, h, r.kernel_name, st["rays"], st["rays"] / n))
    print("per pixel (mean over all / over pixels whose wave walked at all: %.1f %%):" % (100.0 * busy.mean()))
    for name, a in (("group boxes tested by the lane", groups), ("leaf boxes tested by the lane", leaves), ("exact tests by the lane", exact),
                    ("entries stepped through by the wave", wave)):
        print("  %-38s %10.1f %10.1f   max %d" % (name, a.mean(), a[busy].mean() if busy.any() else 0, a.max()))
    own = (groups + leaves).astype(np.float64)
    print("lane's own steps / wave's steps: %.3f (1 = every lane needs every step its wave takes)" % (own[busy].sum() / wave[busy].sum()))
    print("per ray: wave steps %.1f, lane steps %.1f, exact tests %.2f" % (wave.sum() / st["rays"], own.sum() / st["rays"], exact.sum() / st["rays"]))


if __name__ == "__main__":
    main(sys.argv[1:])
