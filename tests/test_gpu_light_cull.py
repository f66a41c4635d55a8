"""Light-cone culling (rtc_kernel_core.h light_cull_mask): objects that no shadow ray of a shade point can reach are
left out of that shade point's area-light samples.  It must never change an answer.  These tests aim at the cull's
decision boundary -- casters grazing the pyramid between shade point and light, casters behind the shade point,
far and near, tiny and huge, planes just above / below / through the light's height range, jitter at the extremes --
and compare every light intensity and every pixel with the oracle, which knows nothing of culling."""
import numpy as np
import pytest

import ray_tracer_challenge_amd as P
from oracle import oracle as O
from ray_tracer_challenge_amd import scenes
from tests import helpers as H

pytestmark = pytest.mark.gpu
f32 = np.float32


def _random_world(rng, n_objects, jitter):
    objs = []
    for k in range(n_objects):
        kind = rng.choice(["sphere", "sphere", "cube", "plane"]) if k else "plane"
        casts = bool(rng.random() < 0.8)
        scale = float(10.0 ** rng.uniform(-1.2, 0.5))
        pos = rng.uniform(-4, 4, 3)
        if kind == "plane":
            t = P.translation(0.0, float(rng.choice([0.0, -1.0, 2.5, 6.0, rng.uniform(-2, 7)])), 0.0)
            if rng.random() < 0.5:
                t = P.chain(t, P.scaling(float(rng.uniform(0.5, 10)), float(rng.choice([0.01, 1.0, -1.0])), 1.0))
            objs.append(P.Plane(t, P.Material(color=tuple(rng.uniform(0.2, 1, 3))), casts_shadow=casts))
        else:
            sx, sy, sz = scale * rng.uniform(0.5, 2.0, 3)
            if rng.random() < 0.15:
                sx = -sx
            t = P.chain(P.translation(*pos), P.scaling(float(sx), float(sy), float(sz)))
            ctor = P.Sphere if kind == "sphere" else P.Cube
            transparent = rng.random() < 0.2
            objs.append(ctor(t, P.Material(color=tuple(rng.uniform(0.2, 1, 3)), reflective=float(rng.choice([0.0, 0.3])),
                                           transparency=0.6 if transparent else 0.0, refractive_index=1.3),
                             casts_shadow=casts))
    corner = rng.uniform(-3, 3, 3) + np.array([0.0, 3.0, 0.0])
    axes = [rng.normal(0, 1, 3) for _ in range(2)]
    if rng.random() < 0.5:  # axis-aligned lights as in the demo
        axes = [np.array([rng.uniform(0.3, 3), 0, 0]), np.array([0, rng.choice([0.0, rng.uniform(0.3, 3)]), rng.uniform(0.3, 3)])]
    u = axes[0] / max(np.linalg.norm(axes[0]), 1e-6) * rng.uniform(0.3, 3)
    v = axes[1] / max(np.linalg.norm(axes[1]), 1e-6) * rng.uniform(0.3, 3)
    light = P.RectangleLight(P.color(1.2, 1.1, 1.0), P.point(*corner), P.vector(*u), int(rng.integers(1, 5)), P.vector(*v),
                             int(rng.integers(1, 5)), jitter)
    return P.World(objs, light)


JITTERS = [("constant", 0.0), ("constant", 1.0), ("constant", 0.5), ("hashed", 99), ("constant", 1.5), ("constant", -0.25)]


@pytest.mark.parametrize("seed", range(12))
def test_intensity_at_matches_oracle_on_random_scenes(seed):
    """Light::intensity_at (the culled sample loop) for 1500 points per scene: on and near object surfaces, in the
    open, behind objects -- every value must be the oracle's, exactly."""
    rng = np.random.default_rng(1000 + seed)
    jitter = JITTERS[seed % len(JITTERS)]
    world = _random_world(rng, int(rng.integers(1, 5)), jitter)
    ow = H.oracle_world(world)
    n = 1500
    pts = rng.uniform(-6, 6, (n, 3))
    pts[: n // 3, 1] = rng.uniform(-0.2, 0.2, n // 3)           # near the usual floor height
    for k, o in enumerate(world.objects):                        # points hugging each object's bounding sphere
        inv = o.transformation_inverse()
        fwd = np.linalg.inv(inv.astype(np.float64))
        d = rng.normal(0, 1, (60, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        local = d * rng.uniform(0.9, 1.3, (60, 1))
        wp = (fwd[:3, :3] @ local.T).T + fwd[:3, 3]
        pts[n // 3 + 60 * k: n // 3 + 60 * (k + 1)] = wp
    pts = np.concatenate([pts, np.ones((n, 1))], axis=1).astype(f32)
    got = world.intensity_at(pts)
    for i in range(n):
        ow.set_pixel(i)
        exp = ow.intensity_at(pts[i])
        assert got[i] == exp, (seed, jitter, i, pts[i], got[i], exp)


@pytest.mark.parametrize("seed", range(10))
def test_random_area_light_scenes_render_like_the_oracle(seed):
    rng = np.random.default_rng(5000 + seed)
    jitter = JITTERS[seed % 4]
    world = _random_world(rng, int(rng.integers(2, 9)), jitter)
    cam = P.Camera(72, 56, scenes.PI / f32(2.5),
                   P.view_transform(P.point(*rng.uniform(-6, 6, 3)) + np.array([0, 2, 0, 0], dtype=f32), P.point(0, 0.5, 0), P.vector(0, 1, 0)))
    for mode in ("0", "1"):
        import os
        os.environ["RTC_AMD_SPECIALIZE"] = mode
        try:
            canvas = cam.render(world, 3)
        finally:
            del os.environ["RTC_AMD_SPECIALIZE"]
        img, rays = H.oracle_camera(cam).render(H.oracle_world(world), 3, threads=8)
        H.assert_images_equal(canvas.data, img, "seed %d specialise=%s" % (seed, mode))
        assert cam.last_stats["rays"] == rays
        assert cam.last_stats["culled_shadow_rays"] <= rays


def test_cull_statistics_are_reported():
    """The demo scene: most floor points see no caster between themselves and the light."""
    world, camera, depth = scenes.soft_shadows(256, 128)
    canvas = camera.render(world, depth)
    st = camera.last_stats
    assert 0 < st["culled_shadow_rays"] < st["rays"]
    img, rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=8)
    H.assert_images_equal(canvas.data, img, "soft_shadows 256x128")
    assert st["rays"] == rays
