#!/usr/bin/env python3
"""development: more seeds for the two secondary fuzz tests (tests/test_gpu_fuzz.py), progress to stdout"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fuzz as T
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in (range(a, b) if len(sys.argv) <= 3 else ()):
    for fn in (T.test_random_rays_through_the_batched_entry_points,) + ((T.test_random_worlds_at_a_size_that_takes_the_default_fast_paths,) if seed % 4 == 0 else ()):
        try:
            fn(seed)
        except AssertionError as e:
            bad.append((seed, fn.__name__, str(e)[:300]))
            print("FAIL", bad[-1], flush=True)
    if seed % 20 == 0:
        print("seed", seed, "failures so far", len(bad), flush=True)
print("done; failures:", bad)


def intensity_sweep(a, b):
    """Light::intensity_at on random points of the fuzz worlds that have an area light, against the oracle."""
    import numpy as np
    import ray_tracer_challenge_amd as P
    from oracle import oracle as O
    bad = 0
    for seed in range(a, b):
        world, _, _ = T._world(seed, P)
        if not hasattr(world.light, "corner"):
            continue
        own, _, _ = T._world(seed, O)
        rng = np.random.default_rng(seed)
        pts = np.concatenate([rng.uniform(-4, 4, (300, 3)), np.ones((300, 1))], axis=1).astype(np.float32)
        got = world.intensity_at(pts)
        for i in range(len(pts)):
            own.set_pixel(i)
            exp = own.intensity_at(pts[i])
            if got[i] != exp:
                bad += 1
                print("FAIL intensity", seed, i, pts[i], got[i], exp, flush=True)
        if seed % 50 == 0:
            print("intensity seed", seed, "failures so far", bad, flush=True)
    print("intensity sweep done; failures:", bad)


if len(sys.argv) > 3 and sys.argv[3] == "intensity":
    intensity_sweep(a, b)
