#!/usr/bin/env python3
"""Wide differential fuzzing on the GPU box: the worlds of tests/wide_worlds.py (log-uniform sizes 1e-3 .. 1e3, scenes up to
1e4 from the origin, thin scalings, horizons, lights touching casters, grazing views) rendered by the HIP path and by the
oracle; every mismatch is written down and, by switching the conservative shortcuts off one at a time, ATTRIBUTED to the
rule that broke (ERROR_BUDGET.md).  Development tool; the seeds it has found live in tests/test_gpu_fuzz_wide.py.

    python tools/fuzz_wide.py --seeds 0:2000 --out gpurun_out/fuzz_wide.jsonl
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SWITCHES = ["RTC_AMD_CELL_CULL", "RTC_AMD_LIGHT_CULL", "RTC_AMD_DARK", "RTC_AMD_FAST_SHADOW", "RTC_AMD_PRUNE", "RTC_AMD_TRI_PRECULL", "RTC_AMD_BVH",
            "RTC_AMD_SCENE_BOX", "RTC_AMD_GATES", "RTC_AMD_CLUSTERS", "RTC_AMD_SCENE_RECT"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:400")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz_wide.jsonl"))
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--threads", type=int, default=min(16, len(os.sched_getaffinity(0))))
    ap.add_argument("--size", default="", help="WxH: render every world at this size instead of its own")
    a = ap.parse_args()
    import ray_tracer_challenge_amd as P
    from oracle import oracle as O
    from ray_tracer_challenge_amd.renderer import Renderer
    from tests import helpers as H
    from tests import wide_worlds as W

    lo, hi = [int(v) for v in a.seeds.split(":")]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    out = open(a.out, "a")
    t0 = time.time()
    bad = 0

    def render(world, camera, depth, env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            r = Renderer(world, camera, device=0)
            imgs = [r.render(depth).cpu().numpy() for _ in range(a.frames)]
            rays = r.stats()["rays"]
            name = r.kernel_name
            r.close()
            return imgs, rays, name
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    def differs(img, exp):
        return ~((img == exp) | (np.isnan(img) & np.isnan(exp)))

    for seed in range(lo, hi):
        rec = {"seed": seed}
        try:
            world, cam, depth, style = W.world(seed, P)
            own, _, _, _ = W.world(seed, O)
            if a.size:
                cam = (int(a.size.split("x")[0]), int(a.size.split("x")[1]), cam[2], cam[3])
            camera = P.Camera(*cam)
            exp, rays = H.oracle_camera(camera).render(own, depth, threads=a.threads)
            rec.update(style=style, size=[cam[0], cam[1]], depth=depth, n_objects=len(world.objects), rays=int(rays),
                       colours=int(len(np.unique(exp.reshape(-1, 3), axis=0))), nan=int(np.isnan(exp).sum()))
            fails = []
            for spec in ("0", "1"):
                imgs, got_rays, name = render(world, camera, depth, {"RTC_AMD_SPECIALIZE": spec})
                for f, img in enumerate(imgs):
                    d = differs(img, exp)
                    if d.any() or got_rays != rays:
                        ys, xs = np.nonzero(d.any(axis=2))
                        fails.append({"spec": spec, "kernel": name, "frame": f, "pixels": int(d.any(axis=2).sum()), "rays": int(got_rays),
                                      "first": [int(xs[0]), int(ys[0])] if len(xs) else None,
                                      "gpu": [float(v) for v in img[ys[0], xs[0]]] if len(xs) else None,
                                      "cpu": [float(v) for v in exp[ys[0], xs[0]]] if len(xs) else None})
                        break
            if fails:
                bad += 1
                rec["fails"] = fails
                spec = fails[0]["spec"]
                cured = []
                for sw in SWITCHES:  # which shortcut, switched off, makes the frame right again?
                    imgs, got_rays, _ = render(world, camera, depth, {"RTC_AMD_SPECIALIZE": spec, sw: "0"})
                    if not differs(imgs[0], exp).any() and got_rays == rays:
                        cured.append(sw)
                rec["cured_by"] = cured
        except Exception as e:  # a world the library refuses, or worse: written down, not fatal
            rec["error"] = "%s: %s" % (type(e).__name__, e)
            bad += 1
        out.write(json.dumps(rec) + "\n")
        out.flush()
        if "fails" in rec or "error" in rec:
            print("seed %d %s: %s" % (seed, rec.get("style"), json.dumps({k: rec[k] for k in ("fails", "cured_by", "error") if k in rec})[:600]), flush=True)
        if (seed - lo) % 100 == 99:
            print("... %d worlds, %d bad, %.0f s" % (seed - lo + 1, bad, time.time() - t0), flush=True)
    print("done: %d worlds, %d bad, %.0f s" % (hi - lo, bad, time.time() - t0))


if __name__ == "__main__":
    main()
