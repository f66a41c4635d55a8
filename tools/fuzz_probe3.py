#!/usr/bin/env python3
"""development: normal and light vector at the hit of one pixel of a fuzz world, oracle against device"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_challenge_amd as P
from oracle import oracle as O
from tests import helpers as H
from tests import test_gpu_fuzz as T
seed, x, y = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
world, cam, depth = T._world(seed, P)
own, _, _ = T._world(seed, O)
camera = P.Camera(640, 420, cam[2], cam[3])
o, d = H.oracle_camera(camera).ray_for_pixel(x, y)
ts, objs = own.intersect(o, d)
t, i = [(t, i) for t, i in zip(ts, objs) if t >= 0][0]
p = o + d * t


def flatten(objs):
    out = []
    for ob in objs:
        if getattr(ob, "is_group", False):
            out += flatten(ob.get_children())
        else:
            out.append(ob)
    return out


ol = flatten(own.objects)
pl = world._c().leaves
print("leaf", int(i), "of", len(ol), len(pl), type(pl[int(i)]).__name__)
n_o = ol[int(i)].normal_at(p)
n_d = pl[int(i)].normal_at(p[None])[0]
print("oracle normal", n_o, n_o.view(np.uint32))
print("device normal", n_d, n_d.view(np.uint32))
print("equal", np.array_equal(n_o, n_d))
print("transform inverse (api side):")
print(np.asarray(pl[int(i)].transform_inverse if hasattr(pl[int(i)], "transform_inverse") else pl[int(i)]._inv if hasattr(pl[int(i)], "_inv") else "?"))
