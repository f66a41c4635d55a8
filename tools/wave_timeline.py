#!/usr/bin/env python3
"""When do a frame's waves run, and for how long?  (development tool)

    python tools/wave_timeline.py [--scene mesh --size 2048 [--height H]] [ENV=value ...]

Renders the scene once with its kernel compiled with -DRTC_DEBUG_TIMELINE: every pixel then holds its wave's start and
end on the 100 MHz clock and its HW_ID instead of a colour.  Prints the number of resident waves over the kernel's
length, the distribution of wave lengths, and where in the image the longest waves are.
"""
import os

# the development switches this tool drives exist only in the development build of the library
os.environ.setdefault("RTC_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ray_tracer_challenge_amd", "librtc_amd_dev.so"))
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(args):
    scene, w, h = "mesh", 2048, 0
    it = iter(args)
    for a in it:
        if a == "--scene": scene = next(it)
        elif a == "--size": w = int(next(it))
        elif a == "--height": h = int(next(it))
        else:
            k, v = a.split("=", 1)
            os.environ[k] = v
    h = h or w
    os.environ["RTC_AMD_JIT_FLAGS"] = (os.environ.get("RTC_AMD_JIT_FLAGS", "") + " -DRTC_DEBUG_TIMELINE").strip()
    os.environ["RTC_AMD_SPECIALIZE"] = "1"
    from ray_tracer_challenge_amd import scenes
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = getattr(scenes, scene)(w, h)
    r = Renderer(world, camera, device=0)
    out = r.alloc()
    for _ in range(3):
        r.render(depth, out=out)
    r.stats()
    r.render(depth, out=out)
    st = r.stats()
    px = out.cpu().numpy().view(np.uint32).reshape(h, w, 3)
    flat = px.reshape(-1, 3)
    written = np.nonzero((flat[:, 0] != 0) | (flat[:, 1] != 0))[0]  # (pixels no wave wrote: the zero-filled part of a scene-rectangle frame)
    flat = flat[written]
    key = flat[:, 0].astype(np.uint64) << np.uint64(32) | flat[:, 2].astype(np.uint64)
    _, first, npix = np.unique(key, return_index=True, return_counts=True)
    start = flat[first, 0].astype(np.int64)
    end = flat[first, 1].astype(np.int64)
    hw = flat[first, 2]
    t0 = start.min()
    start, end = start - t0, end - t0
    end[end < start] += 1 << 32
    dur = end - start
    total = end.max()
    print("%s %dx%d: kernel %.3f ms (events), %d waves seen, span %.3f ms (100 MHz ticks: %d)  kernel %s" % (
        scene, w, h, st["kernel_ms"], len(start), total / 1e5, total, r.kernel_name))
    # resident waves over time, in 20 slices
    n_slices = 20
    edges = np.linspace(0, total, n_slices + 1)
    print("resident waves (mean) per 1/%d of the span; waves started in the slice; mean length of those (us)" % n_slices)
    for i in range(n_slices):
        a, b = edges[i], edges[i + 1]
        overlap = np.clip(np.minimum(end, b) - np.maximum(start, a), 0, None).sum() / max(b - a, 1)
        started = (start >= a) & (start < b)
        print("  %5.2f-%5.2f ms  resident %7.0f  started %6d  mean len %8.1f us  max len %8.1f us" % (
            a / 1e5, b / 1e5, overlap, started.sum(), dur[started].mean() / 100 if started.any() else 0, dur[started].max() / 100 if started.any() else 0))
    q = np.percentile(dur, [50, 90, 99, 99.9, 100]) / 100
    print("wave length us: median %.1f  p90 %.1f  p99 %.1f  p99.9 %.1f  max %.1f;  sum of lengths %.1f wave-ms" % (*q, dur.sum() / 1e5))
    order = np.argsort(-dur)[:12]
    print("longest waves: start ms, length ms, first pixel (x, y), pixels, xcc/se/cu/simd")
    for i in order:
        y, x = divmod(int(written[first[i]]), w)
        v = int(hw[i])
        print("  %.3f  %.3f  (%d, %d)  %d  xcc %d se %d cu %d simd %d" % (start[i] / 1e5, dur[i] / 1e5, x, y, npix[i], v >> 16, (v >> 13) & 7, (v >> 8) & 15, (v >> 4) & 3))
    # per-SIMD busy: sum of wave lengths per (xcc, se, cu, simd)
    simd = (hw >> 16).astype(np.int64) * 4096 + ((hw >> 13) & 7) * 256 + ((hw >> 8) & 15) * 4 + ((hw >> 4) & 3)
    ids, inv = np.unique(simd, return_inverse=True)
    busy = np.bincount(inv, weights=dur.astype(np.float64))
    last = np.zeros(len(ids))
    np.maximum.at(last, inv, end.astype(np.float64))
    print("%d SIMDs seen; wave-time per SIMD / span: min %.2f median %.2f max %.2f;  last wave end per SIMD (ms): min %.3f median %.3f max %.3f" % (
        len(ids), busy.min() / total, np.median(busy) / total, busy.max() / total, last.min() / 1e5, np.median(last) / 1e5, last.max() / 1e5))


if __name__ == "__main__":
    main(sys.argv[1:])
