#!/usr/bin/env python3
"""development: more seeds for the two secondary fuzz tests (tests/test_gpu_fuzz.py), progress to stdout"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fuzz as T
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(a, b):
    for fn in (T.test_random_rays_through_the_batched_entry_points,) + ((T.test_random_worlds_at_a_size_that_takes_the_default_fast_paths,) if seed % 4 == 0 else ()):
        try:
            fn(seed)
        except AssertionError as e:
            bad.append((seed, fn.__name__, str(e)[:300]))
            print("FAIL", bad[-1], flush=True)
    if seed % 20 == 0:
        print("seed", seed, "failures so far", len(bad), flush=True)
print("done; failures:", bad)
