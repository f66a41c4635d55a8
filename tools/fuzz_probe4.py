#!/usr/bin/env python3
"""development: the six shadow rays of one shade point of fuzz world 84, oracle against device"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_challenge_amd as P
from oracle import oracle as O
from tests import helpers as H
from tests import test_gpu_fuzz as T
world, cam, depth = T._world(84, P)
own, _, _ = T._world(84, O)
camera = P.Camera(640, 420, cam[2], cam[3])
o, d = H.oracle_camera(camera).ray_for_pixel(318, 12)
ts, objs = own.intersect(o, d)
c = own.precompute_values(o, d, 0, [(float(ts[0]), int(objs[0]))])
op = np.array(c.over_point[:], dtype=np.float32)
print("over_point", op)
print("intensity_at(over_point): oracle", own.intensity_at(op), "device", world.intensity_at(op[None])[0])
for v in range(2):
    for u in range(3):
        lp = own.point_on_light(u, v)
        so = own.is_shadowed(lp, op)
        sd = world.is_shadowed(lp[None], op[None])[0]
        print("cell", u, v, "light point", lp, "oracle shadowed", so, "device", bool(sd), "" if bool(sd) == so else "<-- DIFFER")
        if bool(sd) != so:
            dirv = lp - op
            dist = np.sqrt((dirv[:3] * dirv[:3]).sum(dtype=np.float32), dtype=np.float32)
            dn = (dirv / dist).astype(np.float32)
            xs = own.intersect(op, dn)
            print("   oracle intersections of the shadow ray:", [(float(t), int(i)) for t, i in zip(*xs)][:10], "distance", dist)
