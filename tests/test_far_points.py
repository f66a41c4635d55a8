"""Shade points thousands of units away (-m gpu): the horizon of a plane.

An f32 quadratic met from a distance D cancels to (D / r)^2 of its terms: what the reference reports for a shadow ray that
starts 4 000 units out and ends beside a small sphere is off by units, and every shortcut whose margin is relative to
coordinates (distance pruning of groups, light-cone culling, the fast shadow decision) must stand back there.  A low camera
looking at the horizon over a floor, small spheres -- loose, in a group, and in a divided group of many -- right at an area
light and at a point light, rotated so that nothing is axis-aligned: every pixel against the oracle.
"""
import os

import numpy as np
import pytest

import ray_tracer_challenge_amd as P
from ray_tracer_challenge_amd import scenes
from tests import helpers as H

pytestmark = pytest.mark.gpu
f32 = np.float32
THREADS = min(16, len(os.sched_getaffinity(0)))


def _world(api, grouping, light_kind, seed):
    rng = np.random.default_rng(seed)
    mat = lambda c, **kw: api.Material(color=c, ambient=0.1, diffuse=0.7, specular=0.3, shininess=50.0, **kw)
    floor = api.Plane(api.rotation_z(f32(0.002)), mat((0.9, 0.9, 0.8), reflective=0.2 if seed % 2 else 0.0))
    balls = []
    for k in range(6 if grouping == "many" else 3):
        r = float(rng.uniform(0.15, 0.5))
        t = api.chain(api.translation(float(rng.uniform(-1.5, 1.5)), float(rng.uniform(1.5, 3.0)), float(rng.uniform(2.0, 5.0))),
                      api.rotation_y(float(rng.uniform(-1, 1))), api.scaling(r, r * float(rng.uniform(0.7, 1.3)), r))
        kind = [api.Sphere, api.Cylinder, api.Cone][k % 3] if grouping != "loose" or k else api.Sphere
        kw = dict(minimum_y=-1.0, maximum_y=1.0, closed=True) if kind is not api.Sphere else {}
        balls.append(kind(t, mat(tuple(rng.uniform(0.2, 1.0, 3))), **kw))
    if grouping == "loose":
        objects = [floor] + balls
    else:
        g = api.GroupShape.with_children(balls)
        if grouping == "many":
            g.divide(2)
        objects = [floor, g]
    if light_kind == "area":
        light = api.RectangleLight(api.color(1.2, 1.2, 1.2), api.point(-0.8, 1.8, 3.0), api.vector(1.6, 0.1, 0.0), 4, api.vector(0.0, 1.2, 0.3), 3,
                                   ("constant", 0.5) if seed % 2 else ("hashed", seed))
    else:
        light = api.PointLight(api.point(0.1, 2.2, 3.4), api.color(1, 1, 1))
    return api.World(objects, light)


@pytest.mark.parametrize("grouping", ["loose", "group", "many"])
@pytest.mark.parametrize("light_kind", ["area", "point"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_the_horizon_of_a_plane(grouping, light_kind, seed):
    from oracle import oracle as O
    from ray_tracer_challenge_amd.renderer import Renderer
    world, own = _world(P, grouping, light_kind, seed), _world(O, grouping, light_kind, seed)
    # the strip of the image around the horizon: floor points from tens to tens of thousands of units away
    camera = P.Camera(640, 96, scenes.PI / f32(12.0), P.view_transform(P.point(0.3, 0.6, -6.0), P.point(0.0, 0.55, 4.0), P.vector(0, 1, 0)))
    exp, rays = H.oracle_camera(camera).render(own, 3, threads=THREADS)
    assert np.isfinite(exp).all()
    r = Renderer(world, camera, device=0)
    for frame in range(3):
        img = r.render(3).cpu().numpy()
        H.assert_images_equal(img, exp, "%s / %s light / seed %d (%s) frame %d" % (grouping, light_kind, seed, r.kernel_name, frame))
        assert r.stats()["rays"] == rays
    r.close()
