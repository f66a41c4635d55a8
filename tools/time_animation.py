#!/usr/bin/env python3
"""An animation: the camera moves a little before every frame (rtc_ctx_set_scene each time).  Kernel ms and wall ms per frame with
the frames scheduled by the frame before (the block lists outlive the change of scene) and without (RTC_AMD_BLOCK_FEEDBACK=0):
    python tools/time_animation.py mesh 2048 2048 [frames] [camera]
`camera`: Renderer.set_camera instead of set_scene -- the world is not flattened again on the Python side (a Python cost, milliseconds
for a mesh: not the library's), and the library keeps the records that are resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ray_tracer_challenge_amd as P
from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.renderer import Renderer
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 16
camera_only = len(sys.argv) > 5 and sys.argv[5] == "camera"
world, camera0, depth = getattr(scenes, name)(w, h)


def camera(frame):
    a = 0.004 * frame  # a quarter of a degree per frame
    return P.Camera(w, h, scenes.PI / np.float32(3.0), P.view_transform(P.point(0.2 + 3.0 * np.sin(a), 2.2, -5.5 * np.cos(a)), P.point(0, 0.8, 0), P.vector(0, 1, 0)))


from ray_tracer_challenge_amd import _lib
if os.environ.get("ANIMATION_DEV_LIB"):  # development switches (RTC_AMD_FEEDBACK_*) live in the development build only
    _lib._lib = _lib.load(_lib.DEV_LIB_PATH)
for mode in ("1", "0", "1", "0"):
    os.environ["RTC_AMD_BLOCK_FEEDBACK"] = mode
    r = Renderer(world, camera(0), device=0)
    out = r.alloc()
    for _ in range(3): r.render(depth, out=out)
    r.stats()
    kernel, wall = [], []
    for f in range(1, frames + 1):
        cam = camera(f)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.set_camera(cam) if camera_only else r.set_scene(world, cam)
        r.render(depth, out=out)
        st = r.stats()
        wall.append((time.perf_counter() - t0) * 1e3)
        kernel.append(st["kernel_ms"])
    r.close()
    if os.environ.get("ANIMATION_SERIES"):
        print("feedback=%s kernel ms per frame:" % mode, " ".join("%.2f" % k for k in kernel), flush=True)
    wl = np.array(wall[4:])
    print("%s %dx%d feedback=%s: kernel ms per frame, frames 5..%d: mean %.3f (min %.3f max %.3f); wall ms incl. %s: median %.2f mean %.2f, the three longest %s (frames %s)" % (
        name, w, h, mode, frames, np.mean(kernel[4:]), np.min(kernel[4:]), np.max(kernel[4:]), "set_camera" if camera_only else "set_scene", np.median(wl), np.mean(wl),
        np.round(np.sort(wl)[-3:], 1).tolist(), (np.argsort(wl)[-3:] + 5).tolist()), flush=True)
