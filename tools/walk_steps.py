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
    # Per WAVE (8 x 8 pixel tiles: one lane per pixel): the entries the wave stepped through -- the union of its lanes' paths, which
    # every lane of the packet walk pays for -- against the entries its BUSIEST lane needed for itself.  union / busiest = 1: the
    # packet costs what the longest path costs alone; 2: the lanes' paths differ enough to double the walk.
    H8, W8 = (h - 1) // 8 * 8, (w - 1) // 8 * 8
    tile = lambda a: a[:H8, :W8].reshape(H8 // 8, 8, W8 // 8, 8)
    union = tile(wave).max(axis=(1, 3)).astype(np.float64)      # (the same number in every lane of the wave)
    busiest = tile(own).max(axis=(1, 3))
    mean_lane = tile(own).mean(axis=(1, 3))
    walked = union > 0
    ratio = union[walked] / np.maximum(busiest[walked], 1.0)
    print("per wave (%d of %d tiles walked): union / busiest lane -- mean %.2f, weighted by the union's size %.2f; union / mean lane %.2f" % (
        walked.sum(), walked.size, ratio.mean(), union[walked].sum() / busiest[walked].sum(), union[walked].sum() / np.maximum(mean_lane[walked].sum(), 1.0)))
    edges = [1.0, 1.1, 1.2, 1.3, 1.5, 1.75, 2.0, 2.5, 3.0, 4.0, 6.0, 1e9]
    hist, _ = np.histogram(ratio, bins=edges)
    wsum = [union[walked][(ratio >= lo) & (ratio < hi)].sum() for lo, hi in zip(edges[:-1], edges[1:])]
    print("  union / busiest   waves      share of waves   share of all wave steps")
    for (lo, hi), n_w, ws in zip(zip(edges[:-1], edges[1:]), hist, wsum):
        print("  %4.2f .. %-8s %8d %14.1f %% %18.1f %%" % (lo, ("%.2f" % hi) if hi < 1e8 else "", n_w, 100.0 * n_w / max(1, walked.sum()), 100.0 * ws / max(1.0, union[walked].sum())))
    # the longest waves decide a frame: the same ratio among the 1 % of waves with the largest unions
    top = union[walked] >= np.quantile(union[walked], 0.99)
    print("  the longest 1 %% of waves (union >= %.0f entries): union / busiest %.2f" % (np.quantile(union[walked], 0.99), union[walked][top].sum() / busiest[walked][top].sum()))


if __name__ == "__main__":
    main(sys.argv[1:])
