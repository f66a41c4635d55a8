#!/usr/bin/env python3
"""Differential fuzzing of the one-call seam (rtc_render_ex through Camera.render, camera.rs:76): the wide worlds of
tests/wide_worlds.py rendered by ONE call each -- f32 rows into a caller's array, then the scale_color'd bytes (canvas.rs:39-43), on
alternating band heights -- against the oracle: every pixel, the ray count, and the bytes against the oracle's own quantisation.
Development tool (the suite's seam tests are tests/test_seam.py).

    python tools/fuzz_seam.py --seeds 0:300 [--size 640x400] --out gpurun_out/fuzz_seam.jsonl
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:300")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz_seam.jsonl"))
    ap.add_argument("--threads", type=int, default=min(16, len(os.sched_getaffinity(0))))
    ap.add_argument("--size", default="", help="WxH: render every world at this size instead of its own (the seam cuts larger frames into reported chunks)")
    a = ap.parse_args()
    import ray_tracer_challenge_amd as P
    from oracle import oracle as O
    from tests import helpers as H
    from tests import wide_worlds as W

    lo, hi = [int(v) for v in a.seeds.split(":")]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    out = open(a.out, "a")
    t0, bad = time.time(), 0
    for seed in range(lo, hi):
        rec = {"seed": seed}
        try:
            world, cam, depth, style = W.world(seed, P)
            own, _, _, _ = W.world(seed, O)
            if a.size:
                cam = (int(a.size.split("x")[0]), int(a.size.split("x")[1]), cam[2], cam[3])
            camera = P.Camera(*cam)
            exp, rays = H.oracle_camera(camera).render(own, depth, threads=a.threads)
            exp8 = O.quantize(exp)  # scale_color as the oracle does it (canvas.rs:39-43)
            rec.update(style=style, size=[cam[0], cam[1]], depth=depth)
            fails = []
            for band in (0, 16, 48):
                got = camera.render(world, depth, band_rows=band).data
                st = dict(camera.last_stats)
                same = (got == exp) | (np.isnan(got) & np.isnan(exp))
                if not same.all() or st["rays"] != rays:
                    fails.append({"what": "f32", "band_rows": band, "pixels": int((~same).any(axis=2).sum()), "rays": st["rays"], "expected_rays": int(rays)})
                got8 = camera.render(world, depth, quantize=True, band_rows=band)
                ok8 = got8 == exp8
                if not ok8.all() or camera.last_stats["rays"] != rays:
                    fails.append({"what": "u8", "band_rows": band, "pixels": int((~ok8).any(axis=2).sum()), "rays": camera.last_stats["rays"]})
            if fails:
                bad += 1
                rec["fails"] = fails
        except Exception as e:  # a world the library refuses, or worse: written down, not fatal
            rec["error"] = "%s: %s" % (type(e).__name__, e)
            bad += 1
        out.write(json.dumps(rec) + "\n")
        out.flush()
        if "fails" in rec or "error" in rec:
            print("seed %d %s: %s" % (seed, rec.get("style"), json.dumps({k: rec[k] for k in ("fails", "error") if k in rec})[:600]), flush=True)
        if (seed - lo) % 50 == 49:
            print("... %d worlds, %d bad, %.0f s" % (seed - lo + 1, bad, time.time() - t0), flush=True)
    print("done: %d worlds, %d bad, %.0f s" % (hi - lo, bad, time.time() - t0))


if __name__ == "__main__":
    main()
