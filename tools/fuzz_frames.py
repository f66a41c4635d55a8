#!/usr/bin/env python3
"""development: the fuzz worlds of tests/test_gpu_fuzz.py at 640x420 with nothing forced, FOUR frames each (a scene's first, and
three scheduled by the frames before), every frame against the oracle:  python tools/fuzz_frames.py 100 220"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fuzz as T
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(a, b):
    try:
        T.test_random_worlds_at_a_size_that_takes_the_default_fast_paths(seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:300]))
        print("FAIL", bad[-1], flush=True)
    if seed % 10 == 0:
        print("seed", seed, "failures so far", len(bad), flush=True)
print("done; failures:", bad)
sys.exit(1 if bad else 0)
