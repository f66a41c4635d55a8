#!/usr/bin/env python3
"""One-off sweep for the margin-guarded fast sample decision (shadow_fast): many random scale+translate-only worlds
(tests/test_gpu_light_cull.py's generator), rendered with RTC_AMD_FAST_SHADOW on and off and by the oracle.
    python tools/fuzz_fast.py [first_seed] [count]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ray_tracer_challenge_amd as P
from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.renderer import Renderer
from tests import helpers as H
from tests.test_gpu_light_cull import _random_simple_world

f32 = np.float32
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 400)
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(7000 + seed)
    jitter = [("hashed", 1234 + seed), ("constant", 0.5), ("constant", 1.0), ("constant", 0.0)][seed % 4]
    world = _random_simple_world(rng, int(rng.integers(2, 9)), jitter)
    cam = P.Camera(88, 66, scenes.PI / f32(2.5),
                   P.view_transform(P.point(*[float(x) for x in rng.uniform(-5, 5, 3) + np.array([0, 3.0, 0])]), P.point(0, 0.7, 0), P.vector(0, 1, 0)))
    exp, rays = H.oracle_camera(cam).render(H.oracle_world(world), 3, threads=8)
    os.environ["RTC_AMD_SPECIALIZE"] = "1" if seed % 2 else "0"
    for fast in ("1", "0"):
        os.environ["RTC_AMD_FAST_SHADOW"] = fast
        r = Renderer(world, cam, device=0)
        img = r.render(3).cpu().numpy()
        st = r.stats()
        r.close()
        if not np.array_equal(img, exp) or st["rays"] != rays:
            bad += 1
            print("MISMATCH seed", seed, "fast", fast, int((img != exp).sum()), st["rays"], rays, flush=True)
    if (seed - first) % 50 == 49:
        print("seed", seed, "done, mismatches so far:", bad, flush=True)
print("swept", count, "seeds from", first, "- mismatches:", bad)
