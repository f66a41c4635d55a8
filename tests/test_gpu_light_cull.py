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
        kind = rng.choice(["sphere", "sphere", "cube", "plane", "cylinder"]) if k else "plane"
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
            if rng.random() < 0.4:  # the cull works in the object's own space under any affine transform
                t = P.chain(t, P.rotation_z(float(rng.uniform(-3, 3))), P.rotation_x(float(rng.uniform(-3, 3))))
            if kind == "cylinder":
                lo = float(rng.uniform(-2, 0))
                objs.append(P.Cylinder(t, P.Material(color=tuple(rng.uniform(0.2, 1, 3))), casts_shadow=casts, minimum_y=lo,
                                       maximum_y=lo + float(rng.uniform(0.2, 3)), closed=bool(rng.random() < 0.5)))
                continue
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
    # 3..5 steps each way: the cull is only attempted for lights of >= 8 cells
    light = P.RectangleLight(P.color(1.2, 1.1, 1.0), P.point(*corner), P.vector(*u), int(rng.integers(3, 7)), P.vector(*v),
                             int(rng.integers(3, 7)), jitter)
    return P.World(objs, light)


JITTERS = [("constant", 0.0), ("constant", 1.0), ("constant", 0.5), ("hashed", 99), ("constant", 1.5), ("constant", -0.25)]


@pytest.mark.parametrize("seed", range(30))
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


@pytest.mark.parametrize("seed", range(20))
def test_random_area_light_scenes_render_like_the_oracle(seed, monkeypatch):
    rng = np.random.default_rng(5000 + seed)
    jitter = JITTERS[seed % 4]
    # seeds 14..19: 10-24 objects, i.e. the any-count loop (and its like-objects specialisation does not apply)
    n_objects = int(rng.integers(10, 25)) if seed >= 14 else int(rng.integers(2, 9))
    world = _random_world(rng, n_objects, jitter)
    cam = P.Camera(72, 56, scenes.PI / f32(2.5),
                   P.view_transform(P.point(*rng.uniform(-6, 6, 3)) + np.array([0, 2, 0, 0], dtype=f32), P.point(0, 0.5, 0), P.vector(0, 1, 0)))
    from ray_tracer_challenge_amd.renderer import Renderer
    img, rays = H.oracle_camera(cam).render(H.oracle_world(world), 3, threads=8)
    for mode in ("0", "1"):
        # a context of its own per mode: the library reads its switches when a context is created, and the kernel family
        # that rendered is checked by name (the one-call seam keeps its context, and would keep its kernel)
        monkeypatch.setenv("RTC_AMD_SPECIALIZE", mode)
        r = Renderer(world, cam, device=0)
        if mode == "0":
            assert r.kernel_name.startswith("render_kernel<"), r.kernel_name
        elif n_objects <= 8:
            assert r.kernel_name.startswith("render_kernel_spec["), r.kernel_name
        got = r.render(3).cpu().numpy()
        st = r.stats()
        r.close()
        H.assert_images_equal(got, img, "seed %d specialise=%s" % (seed, mode))
        assert st["rays"] == rays
        assert st["culled_shadow_rays"] <= rays
    # ... and the one-call seam, which must follow the switch too (the Python mirror drops the seam's contexts when the
    # RTC_AMD_* environment changes; a C caller uses rtc_render_release)
    monkeypatch.setenv("RTC_AMD_SPECIALIZE", "1" if seed % 2 else "0")
    canvas = cam.render(world, 3)
    H.assert_images_equal(canvas.data, img, "seed %d, one call" % seed)
    assert cam.last_stats["rays"] == rays


def test_cull_statistics_are_reported():
    """The demo scene: most floor points see no caster between themselves and the light."""
    world, camera, depth = scenes.soft_shadows(256, 128)
    canvas = camera.render(world, depth)
    st = camera.last_stats
    assert 0 < st["culled_shadow_rays"] < st["rays"]
    img, rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=8)
    H.assert_images_equal(canvas.data, img, "soft_shadows 256x128")
    assert st["rays"] == rays


@pytest.mark.parametrize("jitter", [("hashed", 3), ("constant", 0.0), ("constant", 1.0)])
def test_dense_sweep_across_shadow_edges(jitter):
    """Shade points in fine steps along lines that run from open floor, through the penumbra, under the objects and
    out again -- so every wave-sized run of points crosses from 'culled' to 'tested' somewhere near the objects'
    silhouettes as seen from the light -- for a sphere, a squashed sphere, a cube and a second, non-casting sphere."""
    floor = P.Plane(None, P.Material())
    ball = P.Sphere(P.translation(0.0, 1.0, 0.0), P.Material())
    disc = P.Sphere(P.chain(P.translation(2.5, 0.6, 0.3), P.scaling(1.2, 0.15, 0.8)), P.Material())
    box = P.Cube(P.chain(P.translation(-2.6, 0.5, -0.2), P.scaling(0.5, 0.5, 0.7)), P.Material())
    ghost = P.Sphere(P.chain(P.translation(0.8, 2.2, 0.1), P.scaling(0.4, 0.4, 0.4)), P.Material(), casts_shadow=False)
    light = P.RectangleLight(P.color(1, 1, 1), P.point(-1.0, 5.0, -1.0), P.vector(2, 0, 0), 4, P.vector(0, 0.5, 2), 3, jitter)
    for objs in ([floor, ball, disc, box], [floor, ball, ghost, box], [ball, disc], [floor, disc, ghost]):
        world = P.World(objs, light)
        ow = H.oracle_world(world)
        xs = np.arange(-6.0, 6.0, 0.004, dtype=np.float64)
        for z, y in ((0.0, 1.19e-3), (0.65, 1.19e-3), (-0.2, 0.3)):
            pts = np.stack([xs, np.full_like(xs, y), np.full_like(xs, z), np.ones_like(xs)], axis=1).astype(f32)
            got = world.intensity_at(pts)
            for i in range(pts.shape[0]):
                ow.set_pixel(i)
                exp = ow.intensity_at(pts[i])
                assert got[i] == exp, (jitter, len(objs), z, pts[i], got[i], exp)
            assert 0.0 < got.mean() < 1.0 or len(objs) == 2


@pytest.mark.parametrize("jitter", [("hashed", 5), ("constant", 0.5)])
def test_non_casters_in_front_of_and_behind_casters(jitter):
    """A non-caster matters to a shadow ray only when it is hit before the nearest caster (world.rs:104-119 takes the
    nearest hit of ALL objects and asks whether it casts).  The kernel leaves a non-caster out when its bounding sphere
    lies behind every caster still in play as seen from the shade point: ghosts -- spheres, a slab like the demo's
    lampshade, a bounded cylinder -- slide along the line from the floor to the light, through and past the casters."""
    light = P.RectangleLight(P.color(1, 1, 1), P.point(-1.0, 6.0, -1.0), P.vector(2, 0, 0), 4, P.vector(0, 0, 2), 4, jitter)
    floor = P.Plane(None, P.Material())
    ball = P.Sphere(P.chain(P.translation(0.0, 2.0, 0.0), P.scaling(0.8, 0.8, 0.8)), P.Material())
    box = P.Cube(P.chain(P.translation(1.6, 1.2, 0.4), P.scaling(0.4, 0.6, 0.4), P.rotation_y(f32(0.5))), P.Material())
    xs = np.arange(-3.0, 3.0, 0.01, dtype=np.float64)
    pts = np.stack([xs, np.full_like(xs, 1.19e-3), np.full_like(xs, 0.1), np.ones_like(xs)], axis=1).astype(f32)
    lit_fraction = []
    for h in (0.6, 1.3, 2.0, 2.9, 3.4, 4.5, 5.9, 6.5):  # below, inside, just above the casters; at and beyond the light
        ghosts = [P.Sphere(P.chain(P.translation(0.1, h, 0.0), P.scaling(1.0, 0.3, 1.0)), P.Material(), casts_shadow=False),
                  P.Cube(P.chain(P.translation(0.0, h + 0.2, 0.0), P.scaling(1.5, 0.01, 1.5)), P.Material(), casts_shadow=False),
                  P.Cylinder(P.chain(P.translation(1.5, h, 0.3), P.scaling(0.5, 1.0, 0.5)), P.Material(), casts_shadow=False,
                             minimum_y=-0.2, maximum_y=0.2, closed=True)]
        for objs in ([floor, ball, box] + ghosts, [ball, ghosts[0], box, ghosts[1]]):
            world = P.World(objs, light)
            ow = H.oracle_world(world)
            got = world.intensity_at(pts)
            for i in range(pts.shape[0]):
                ow.set_pixel(i)
                exp = ow.intensity_at(pts[i])
                assert got[i] == exp, (jitter, h, len(objs), pts[i], got[i], exp)
            lit_fraction.append(float(got.mean()))
    assert min(lit_fraction) < 0.9 and max(lit_fraction) > min(lit_fraction) + 0.02  # the ghosts do change the answer when in front


@pytest.mark.parametrize("jitter", [("hashed", 8), ("constant", 1.0)])
def test_points_on_the_far_side_of_a_casting_sphere(jitter):
    """'Every sample blocked' is answered without a test for shade points just outside a casting sphere whose whole
    light pyramid points into it.  Points at heights 1e-6 .. 0.3 radii all around spheres (round, squashed, rotated,
    mirrored), with and without a non-caster that may be hit first, against the oracle."""
    rng = np.random.default_rng(17)
    light = P.RectangleLight(P.color(1, 1, 1), P.point(-1.5, 5.0, 3.0), P.vector(3, 0, 0), 5, P.vector(0, 2, 1), 4, jitter)
    for k in range(6):
        t = P.chain(P.translation(*rng.uniform(-1, 1, 3)), P.rotation_z(float(rng.uniform(-3, 3))),
                    P.scaling(*[float(s) for s in rng.uniform(0.3, 1.5, 3) * rng.choice([1.0, 1.0, -1.0], 3)]))
        ball = P.Sphere(t, P.Material())
        objs = [ball, P.Plane(P.translation(0.0, -3.0, 0.0), P.Material())]
        if k % 3 == 1:   # a non-caster hugging the ball: may be hit first, the closed form must stand down
            objs.append(P.Sphere(P.chain(t, P.scaling(1.05, 1.05, 1.05)), P.Material(), casts_shadow=False))
        if k % 3 == 2:   # a non-caster far behind: left out, the closed form applies
            objs.append(P.Cube(P.chain(P.translation(0.0, 5.5, 3.5), P.scaling(2.0, 1.5, 0.01)), P.Material(), casts_shadow=False))
        n = 4000
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        h = 10.0 ** rng.uniform(-6, -0.5, (n, 1))
        local = np.concatenate([d * (1.0 + h), np.ones((n, 1))], axis=1)
        pts = (np.asarray(t, dtype=np.float64) @ local.T).T.astype(f32)
        world = P.World(objs, light)
        ow = H.oracle_world(world)
        got = world.intensity_at(pts)
        for i in range(n):
            ow.set_pixel(i)
            exp = ow.intensity_at(pts[i])
            assert got[i] == exp, (jitter, k, pts[i], got[i], exp)
        assert (got == 0.0).mean() > 0.2 and (got == 1.0).mean() > 0.2


@pytest.mark.parametrize("jitter", [("hashed", 12), ("constant", 0.0)])
def test_points_all_around_cubes_and_cylinders(jitter):
    """A cube or cylinder is left out for shade points beyond one of its faces (or radially outside) whose samples all move
    further out -- in particular points on the object itself facing the light.  Points 1e-6 .. 0.3 units off every face,
    wall and cap of rotated, squashed, mirrored cubes and open / closed cylinders, against the oracle."""
    rng = np.random.default_rng(23)
    light = P.RectangleLight(P.color(1, 1, 1), P.point(-2.0, 5.0, -3.0), P.vector(3, 0, 1), 4, P.vector(0, 2, 0), 3, jitter)
    floor = P.Plane(P.translation(0.0, -4.0, 0.0), P.Material())
    for k in range(8):
        t = P.chain(P.translation(*rng.uniform(-1, 1, 3)), P.rotation_y(float(rng.uniform(-3, 3))), P.rotation_x(float(rng.uniform(-1, 1))),
                    P.scaling(*[float(s) for s in rng.uniform(0.4, 1.5, 3) * rng.choice([1.0, 1.0, -1.0], 3)]))
        n = 3000
        h = 10.0 ** rng.uniform(-6, -0.5, n)
        if k % 2 == 0:
            shape = P.Cube(t, P.Material())
            q = rng.uniform(-1, 1, (n, 3))
            axis = rng.integers(0, 3, n)
            side = rng.choice([-1.0, 1.0], n)
            q[np.arange(n), axis] = side * (1.0 + h)
        else:
            lo, hi = float(rng.uniform(-2, -0.2)), float(rng.uniform(0.2, 2))
            shape = P.Cylinder(t, P.Material(), minimum_y=lo, maximum_y=hi, closed=bool(k % 4 == 1))
            ang = rng.uniform(0, 2 * np.pi, n)
            rad = np.where(rng.random(n) < 0.6, 1.0 + h, rng.uniform(0, 1, n))           # wall, or over / under a cap
            y = np.where(rad > 1.0, rng.uniform(lo, hi, n), np.where(rng.random(n) < 0.5, hi + h, lo - h))
            q = np.stack([rad * np.cos(ang), y, rad * np.sin(ang)], axis=1)
        local = np.concatenate([q, np.ones((n, 1))], axis=1)
        pts = (np.asarray(t, dtype=np.float64) @ local.T).T.astype(f32)
        objs = [shape, floor] + ([P.Sphere(P.chain(P.translation(0.0, 2.5, -1.0), P.scaling(0.5, 0.5, 0.5)), P.Material())] if k % 3 == 0 else [])
        world = P.World(objs, light)
        ow = H.oracle_world(world)
        got = world.intensity_at(pts)
        for i in range(n):
            ow.set_pixel(i)
            exp = ow.intensity_at(pts[i])
            assert got[i] == exp, (jitter, k, pts[i], got[i], exp)
        assert (got == 1.0).mean() > 0.03 and (got < 1.0).mean() > 0.03  # lit and shadowed points both occur


def _random_simple_world(rng, n_objects, jitter):
    """Scale+translate-only spheres and planes (what the SIMPLE kernels and their margin-guarded fast sample decision,
    shadow_fast, are for): casters and non-casters, tiny and huge, mirrored, touching and overlapping, a floor."""
    objs = [P.Plane(P.translation(0.0, float(rng.choice([0.0, -0.5, rng.uniform(-1, 0.5)])), 0.0),
                    P.Material(color=tuple(rng.uniform(0.3, 1, 3)), specular=float(rng.choice([0.0, 0.4])), reflective=float(rng.choice([0.0, 0.25]))),
                    casts_shadow=bool(rng.random() < 0.7))]
    for _ in range(n_objects - 1):
        if rng.random() < 0.15:
            t = P.chain(P.translation(0.0, float(rng.uniform(2.0, 7.0)), 0.0), P.scaling(1.0, float(rng.choice([1.0, -1.0, 0.2])), 1.0))
            objs.append(P.Plane(t, P.Material(color=tuple(rng.uniform(0.3, 1, 3))), casts_shadow=bool(rng.random() < 0.5)))
            continue
        s = float(10.0 ** rng.uniform(-1.3, 0.3)) * rng.uniform(0.6, 1.6, 3)
        if rng.random() < 0.4:
            s[:] = s[0]  # a uniformly scaled sphere: the fast decision factors the scale out (SHAPE_UNIFORM)
        if rng.random() < 0.2:
            s[int(rng.integers(0, 3))] *= -1.0
        pos = rng.uniform(-2.5, 2.5, 3) + np.array([0.0, 1.0, 0.0])
        objs.append(P.Sphere(P.chain(P.translation(*[float(v) for v in pos]), P.scaling(*[float(v) for v in s])),
                             P.Material(color=tuple(rng.uniform(0.2, 1, 3)), specular=float(rng.choice([0.0, 0.0, 0.6])),
                                        reflective=float(rng.choice([0.0, 0.0, 0.3]))),
                             casts_shadow=bool(rng.random() < 0.8)))
    corner = rng.uniform(-2.5, 2.5, 3) + np.array([0.0, 3.5, 0.0])
    u = np.array([rng.uniform(0.2, 3), 0.0, rng.uniform(-0.5, 0.5)])
    v = np.array([0.0, float(rng.choice([0.0, rng.uniform(0.2, 2)])), rng.uniform(0.2, 3)])
    light = P.RectangleLight(P.color(1.3, 1.2, 1.1), P.point(*corner), P.vector(*u), int(rng.integers(3, 9)), P.vector(*v), int(rng.integers(3, 9)), jitter)
    return P.World(objs, light)


@pytest.mark.parametrize("seed", range(24))
def test_fast_sample_decision_changes_nothing_on_random_simple_scenes(seed, monkeypatch):
    """shadow_fast decides an area-light sample from the unnormalised ray wherever its margins hold and every object in
    play is a casting sphere or plane.  Random scale+translate-only worlds, ahead-of-time and scene-compiled kernels:
    images, ray counts and shaded hits with it on equal those with it off -- and the oracle's."""
    from ray_tracer_challenge_amd.renderer import Renderer
    rng = np.random.default_rng(7000 + seed)
    jitter = [("hashed", 1234 + seed), ("constant", 0.5), ("constant", 1.0), ("constant", 0.0)][seed % 4]
    world = _random_simple_world(rng, int(rng.integers(2, 9)), jitter)
    cam = P.Camera(88, 66, scenes.PI / f32(2.5),
                   P.view_transform(P.point(*[float(x) for x in rng.uniform(-5, 5, 3) + np.array([0, 3.0, 0])]), P.point(0, 0.7, 0), P.vector(0, 1, 0)))
    exp, rays = H.oracle_camera(cam).render(H.oracle_world(world), 3, threads=8)
    for spec in ("0", "1"):
        monkeypatch.setenv("RTC_AMD_SPECIALIZE", spec)
        frames = {}
        for fast in ("1", "0"):
            monkeypatch.setenv("RTC_AMD_FAST_SHADOW", fast)
            r = Renderer(world, cam, device=0)
            assert "simple" in r.kernel_name, r.kernel_name
            frames[fast] = (r.render(3).cpu().numpy(), r.stats())
            r.close()
        H.assert_images_equal(frames["1"][0], exp, "seed %d specialise=%s fast" % (seed, spec))
        H.assert_images_equal(frames["0"][0], exp, "seed %d specialise=%s exact" % (seed, spec))
        for key in ("rays", "shaded_hits"):
            assert frames["1"][1][key] == frames["0"][1][key], (seed, spec, key)
        assert frames["1"][1]["rays"] == rays


@pytest.mark.parametrize("seed", range(12))
def test_block_cones_call_blocks_lit_and_change_nothing(seed, monkeypatch):
    """ERROR_BUDGET.md B10: for shade points the whole-light cull leaves, 2 x 2 blocks of the light's cells whose cone misses every casting
    sphere are called lit by the wave without a sample.  Worlds of uniformly scaled casting spheres over a floor under lights with even
    step counts (the only ones the blocks apply to), tiny to huge, near and far: the image and the ray count are the oracle's with the
    blocks on and off, and with them on more rays are answered without a sample."""
    from ray_tracer_challenge_amd.renderer import Renderer
    rng = np.random.default_rng(900 + seed)
    S = float(10.0 ** rng.uniform(-1.5, 1.5))
    objs = [P.Plane(P.translation(0.0, float(-1.0 * S), 0.0), P.Material(color=(0.9, 0.9, 0.8), specular=0.0))]
    for _ in range(int(rng.integers(1, 4))):
        r = S * float(10.0 ** rng.uniform(-1.0, 0.0))
        if rng.random() < 0.15:
            r = -r  # a mirrored sphere: the sign of g
        objs.append(P.Sphere(P.chain(P.translation(*[float(v) for v in rng.uniform(-2, 2, 3) * S]), P.scaling(r, r, r)),
                             P.Material(color=tuple(rng.uniform(0.2, 1, 3)), reflective=float(rng.choice([0.0, 0.4])), specular=0.0)))
    us, vs = [(4, 4), (10, 10), (2, 8), (6, 4)][seed % 4]
    if rng.random() < 0.5:
        u, v = np.array([rng.uniform(0.5, 3) * S, 0, 0]), np.array([0, 0, rng.uniform(0.5, 3) * S])
    else:
        u = rng.normal(size=3)
        v = np.cross(u, rng.normal(size=3))
        u, v = u / np.linalg.norm(u) * rng.uniform(0.5, 3) * S, v / np.linalg.norm(v) * rng.uniform(0.5, 3) * S
    light = P.RectangleLight(P.color(1.2, 1.1, 1.0), P.point(*[float(x) for x in (rng.uniform(-2, 2, 3) + np.array([0, 5, 0])) * S]), P.vector(*[float(x) for x in u]), us,
                             P.vector(*[float(x) for x in v]), vs, ("hashed", seed) if seed % 3 else ("constant", float(rng.choice([0.0, 0.5, 1.0]))))
    world = P.World(objs, light)
    far = float(10.0 ** rng.uniform(0.7, 2.2))
    cam = P.Camera(120, 88, float(min(1.2, 6.0 / far * 2.0)), P.view_transform(P.point(*[float(x) for x in (np.array([0.3, 0.6, -1.0]) * far * S)]), P.point(0, 0, 0), P.vector(0, 1, 0)))
    exp, rays = H.oracle_camera(cam).render(H.oracle_world(world), 3, threads=8)
    culled = {}
    for cells in ("1", "0"):
        for spec in ("0", "1"):
            monkeypatch.setenv("RTC_AMD_CELL_CULL", cells)
            monkeypatch.setenv("RTC_AMD_SPECIALIZE", spec)
            r = Renderer(world, cam, device=0)
            got = r.render(3).cpu().numpy()
            st = r.stats()
            r.close()
            H.assert_images_equal(got, exp, "seed %d blocks %s specialise %s" % (seed, cells, spec))
            assert st["rays"] == rays
            culled[(cells, spec)] = st["culled_shadow_rays"]
    assert culled[("1", "1")] >= culled[("0", "1")] and culled[("1", "0")] >= culled[("0", "0")], culled

