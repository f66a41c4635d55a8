#!/usr/bin/env python3
"""development: the fuzz worlds of tests/test_gpu_fuzz.py at 640x420 seen by two cameras in turn (rtc_ctx_set_scene between
frames: A A A B B A B B B A), every frame against the oracle -- block lists that outlive a change of scene must not change a pixel:
    python tools/fuzz_animation.py 0 100"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_challenge_amd as P
from oracle import oracle as O
from ray_tracer_challenge_amd.renderer import Renderer
from tests import helpers as H
from tests import test_gpu_fuzz as T
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(a, b):
    world, cam, depth = T._world(seed, P)
    own, _, _ = T._world(seed, O)
    shift = P.translation(0.15, -0.1, 0.2)
    cams = [P.Camera(640, 420, cam[2], cam[3]), P.Camera(640, 420, cam[2], P.chain(cam[3], shift))]
    exp = [H.oracle_camera(c).render(own, depth, threads=8) for c in cams]
    r = Renderer(world, cams[0], device=0)
    try:
        for frame, k in enumerate((0, 0, 0, 1, 1, 0, 1, 1, 1, 0)):
            r.set_scene(world, cams[k])
            img = r.render(depth).cpu().numpy()
            if not (np.array_equal(img.view(np.uint32), exp[k][0].view(np.uint32)) or bool(((img == exp[k][0]) | (np.isnan(img) & np.isnan(exp[k][0]))).all())):
                raise AssertionError("pixels differ in frame %d (camera %d)" % (frame, k))
            if r.stats()["rays"] != exp[k][1]:
                raise AssertionError("ray count differs in frame %d (camera %d)" % (frame, k))
    except AssertionError as e:
        bad.append((seed, r.kernel_name, str(e)))
        print("FAIL", bad[-1], flush=True)
    r.close()
    if seed % 10 == 0:
        print("seed", seed, "failures so far", len(bad), flush=True)
print("done; failures:", bad)
sys.exit(1 if bad else 0)
