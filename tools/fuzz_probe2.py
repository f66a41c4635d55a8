#!/usr/bin/env python3
"""development: the differing pixel of a fuzz world with parts of the materials switched off on both sides"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_challenge_amd as P
from oracle import oracle as O
from tests import helpers as H
from tests import test_gpu_fuzz as T
seed, x, y = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
orig_material, orig_pattern = T._material, T._pattern
for what in ("as is", "no specular", "no pattern", "no specular, no pattern", "no diffuse"):
    def material(Px, rng, what=what):
        m = orig_material(Px, rng)
        kw = dict(color=m.color, ambient=m.ambient, diffuse=m.diffuse, specular=m.specular, shininess=m.shininess, reflective=m.reflective,
                  transparency=m.transparency, refractive_index=m.refractive_index, pattern=m.pattern)
        if "no specular" in what: kw["specular"] = 0.0
        if "no pattern" in what: kw["pattern"] = None
        if "no diffuse" in what: kw["diffuse"] = 0.0
        return Px.Material(**kw)
    T._material = material
    world, cam, depth = T._world(seed, P)
    own, _, _ = T._world(seed, O)
    camera = P.Camera(640, 420, cam[2], cam[3])
    o, d = H.oracle_camera(camera).ray_for_pixel(x, y)
    a, b = own.color_at(o, d, depth), world.color_at(o[None], d[None], depth)[0]
    print("%-26s oracle %s device %s %s" % (what, a, b, "EQUAL" if np.array_equal(a, b) else "differ"), flush=True)
T._material = orig_material
