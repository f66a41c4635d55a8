#!/usr/bin/env python3
"""development: one pixel of a fuzz world, stage by stage, device against oracle:  python tools/fuzz_probe.py 84 318 12"""
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
oc = H.oracle_camera(camera)
o, d = oc.ray_for_pixel(x, y)
po, pd = camera.ray_for_pixel(x, y)
print("ray oracle", o, d, "api", po, pd, "equal", np.array_equal(o, po) and np.array_equal(d, pd))
own.set_pixel(y * 640 + x) if hasattr(own, "set_pixel") else None
ts, objs = own.intersect(o, d)
print("oracle intersections (t, obj):", [(float(t), int(i)) for t, i in zip(ts, objs)][:12], "depth", depth)
print("oracle color_at", own.color_at(o, d, depth), "device color_at", world.color_at(o[None], d[None], depth)[0])
hit = [(t, i) for t, i in zip(ts, objs) if t >= 0]
if hit:
    t, i = hit[0]
    p = o + d * t
    print("hit t %r obj %d point %s" % (float(t), int(i), p))
    print("oracle intensity_at(point)", own.intensity_at(p), "device", world.intensity_at(p[None])[0])
    lp = own.light_info()[0]
    print("light", own.light_info())
if hit:
    leaves = world._c().leaves if hasattr(world._c(), "leaves") else None
    shape = leaves[int(i)] if leaves is not None else None
    print("object", int(i), type(shape).__name__ if shape is not None else None)
    if shape is not None:
        m = shape.material
        print("material color", m.color, "ambient", m.ambient, "diffuse", m.diffuse, "specular", m.specular, "shininess", m.shininess, "pattern", type(m.pattern).__name__ if m.pattern is not None else None)
        # over_point as the renderers form it needs the normal; compare the pattern at the hit point itself and nearby
        if m.pattern is not None:
            oleaves = own._leaves() if hasattr(own, "_leaves") else None
            for q in (p, p + np.array([0, 1e-3, 0, 0], dtype=np.float32)):
                try:
                    dv = m.pattern.color_at_object(q[None], shape)[0]
                except Exception as e:
                    dv = repr(e)
                print("  device pattern at", q[:3], dv)
        print("device normal_at", shape.normal_at(p[None])[0] if hasattr(shape, "normal_at") else None)
