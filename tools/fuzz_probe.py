#!/usr/bin/env python3
"""development: one pixel of a fuzz world (tests/test_gpu_fuzz.py, 640x420) stage by stage, device against oracle -- the ray, the
oracle's intersections, color_at, the hit's precomputed values, intensity_at(over_point) and every shadow ray of an area light:
    python tools/fuzz_probe.py 84 318 12
(how the one-pixel difference of world 84 was traced to a shadow ray that ends beside a small sphere, LABNOTES round 3)"""
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
print("ray", o, d, "depth", depth)
print("oracle intersections (t, object):", [(float(t), int(i)) for t, i in zip(ts, objs)][:12])
print("color_at: oracle", own.color_at(o, d, depth), "device", world.color_at(o[None], d[None], depth)[0])
hits = [k for k, t in enumerate(ts) if t >= 0]
if hits:
    c = own.precompute_values(o, d, hits[0], [(float(t), int(i)) for t, i in zip(ts, objs)])
    for k, _t in c._fields_:
        v = getattr(c, k)
        print("  comps.%s" % k, np.array(v[:]) if hasattr(v, "__len__") else v)
    op = np.array(c.over_point[:], dtype=np.float32)
    print("intensity_at(over_point): oracle", own.intensity_at(op), "device", world.intensity_at(op[None])[0])
    pos, u, v, cells = own.light_info()
    if cells > 1:
        us, vs = world.light.u_steps, world.light.v_steps
        for vv in range(vs):
            for uu in range(us):
                lp = own.point_on_light(uu, vv)
                so, sd = own.is_shadowed(lp, op), bool(world.is_shadowed(lp[None], op[None])[0])
                print("  cell", uu, vv, "oracle shadowed", so, "device", sd, "" if sd == so else "<-- DIFFER")
                if sd != so:
                    dirv = lp - op
                    dist = np.sqrt((dirv[:3] * dirv[:3]).sum(dtype=np.float32), dtype=np.float32)
                    xs = own.intersect(op, (dirv / dist).astype(np.float32))
                    print("     oracle intersections of that shadow ray:", [(float(t), int(i)) for t, i in zip(*xs)][:10], "light at", dist)
