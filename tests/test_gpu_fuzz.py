"""Differential fuzzing of the whole render path: random worlds that mix everything the path supports -- all six shape
kinds under arbitrary affine transforms, nested and divided GroupShapes, small parsed meshes, procedural patterns and
UV texture maps, point and area lights with either jitter source, non-casters, mirrors and glass -- rendered by
whichever kernel family and shortcuts the library picks for them (unrolled / any-count / traversal kernels, group
gates, the library's own hierarchy, light-cone culling, triangle pre-culling, several lanes per pixel, ...) and compared
with the oracle, which has none of those.  Bit-exact images, equal ray counts."""
import os

import numpy as np
import pytest

import ray_tracer_challenge_amd as P
from oracle import oracle as O
from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.obj_parser import parse_obj
from ray_tracer_challenge_amd.renderer import Renderer
from tests import helpers as H

pytestmark = pytest.mark.gpu
f32 = np.float32


def _transform(P, rng, spread=3.0, smin=0.25, smax=1.2):
    t = P.translation(*[float(v) for v in rng.uniform(-spread, spread, 3)])
    if rng.random() < 0.6:
        t = P.chain(t, P.rotation_y(float(rng.uniform(-3, 3))), P.rotation_x(float(rng.uniform(-1, 1))))
    s = rng.uniform(smin, smax, 3) if rng.random() < 0.5 else np.full(3, rng.uniform(smin, smax))
    if rng.random() < 0.1:
        s[int(rng.integers(0, 3))] *= -1.0
    t = P.chain(t, P.scaling(*[float(v) for v in s]))
    if rng.random() < 0.1:
        t = P.chain(t, P.shearing(float(rng.uniform(-0.3, 0.3)), 0.0, 0.0, float(rng.uniform(-0.3, 0.3)), 0.0, 0.0))
    return t


def _pattern(P, rng):
    a, b = tuple(rng.uniform(0, 1, 3)), tuple(rng.uniform(0, 1, 3))
    pt = P.chain(P.rotation_z(float(rng.uniform(-1, 1))), P.scaling(*[float(v) for v in rng.uniform(0.1, 0.6, 3)]))
    k = int(rng.integers(0, 8))
    if k < 5:
        return [P.Stripes, P.Gradient, P.Rings, P.Checkers, P.Sine2D][k](a, b, pt)
    if k == 5:
        return P.TextureMap(P.UVCheckers(float(rng.integers(2, 9)), float(rng.integers(2, 9)), a, b),
                            [P.SphericalMap, P.PlanarMap, P.CylindricalMap][int(rng.integers(0, 3))]())
    if k == 6:
        return P.TextureMap(P.UVImage(P.canvas_from_ppm(scenes.synthetic_ppm(16, 8, seed=int(rng.integers(1, 99))))), P.SphericalMap())
    return scenes.align_check_cubic_map(P)


def _material(P, rng):
    u = rng.random()
    return P.Material(color=tuple(rng.uniform(0.1, 1.0, 3)), ambient=float(rng.uniform(0.05, 0.3)), diffuse=float(rng.uniform(0.4, 0.9)),
                      specular=float(rng.choice([0.0, 0.3, 0.9])), shininess=float(rng.choice([10.0, 50.0, 200.0])),
                      reflective=float(rng.uniform(0.2, 0.9)) if u < 0.25 else 0.0,
                      transparency=float(rng.uniform(0.4, 0.95)) if 0.25 <= u < 0.45 else 0.0,
                      refractive_index=float(rng.choice([1.0, 1.33, 1.5, 2.4])), pattern=_pattern(P, rng) if rng.random() < 0.3 else None)


def _leaf(P, rng):
    kind = rng.choice(["sphere", "sphere", "cube", "cylinder", "cone", "triangle"])
    casts = bool(rng.random() < 0.85)
    m, t = _material(P, rng), _transform(P, rng)
    if rng.random() < 0.04:  # a plane anywhere, also inside groups (whose boxes then hold infinities and NaNs)
        return P.Plane(t, m, casts_shadow=casts)
    if kind == "sphere":
        return P.Sphere(t, m, casts_shadow=casts)
    if kind == "cube":
        return P.Cube(t, m, casts_shadow=casts)
    if kind in ("cylinder", "cone"):
        lo = float(rng.uniform(-1.5, 0.0))
        kw = dict(minimum_y=lo, maximum_y=lo + float(rng.uniform(0.3, 2.0)), closed=bool(rng.random() < 0.6)) if rng.random() < 0.85 else {}
        return (P.Cylinder if kind == "cylinder" else P.Cone)(t, m, casts_shadow=casts, **kw)
    pts = [P.point(*[float(v) for v in rng.uniform(-1.5, 1.5, 3)]) for _ in range(3)]
    return P.Triangle(*pts, t, m, casts_shadow=casts)


def _group(P, rng, depth, budget):
    g = P.GroupShape()
    if rng.random() < 0.7:
        g.set_transformation(_transform(P, rng, spread=1.5, smin=0.6, smax=1.3))
    for _ in range(int(rng.integers(1, 6))):
        if budget[0] <= 0:
            break
        if depth < 2 and rng.random() < 0.3:
            g.add_child(_group(P, rng, depth + 1, budget))
        else:
            budget[0] -= 1
            g.add_child(_leaf(P, rng))
    if rng.random() < 0.5:
        g.divide(int(rng.integers(1, 4)))
    return g


def _mesh(P, rng):
    text = scenes.bumpy_mesh_obj(int(rng.integers(5, 9)), int(rng.integers(4, 7)), bool(rng.random() < 0.5))
    g = parse_obj(text, P).take_all_as_group()
    g.set_material(_material(P, rng))
    g.set_transformation(_transform(P, rng, spread=2.0, smin=0.5, smax=1.0))
    g.divide(int(rng.integers(2, 7)))
    return g


def _world(seed, P):
    rng = np.random.default_rng(1000 + seed)
    # 0: few flat objects; 1: many flat spheres / cubes; 2: small tree; 3: larger trees + mesh; 4: anything;
    # 5: big overlapping glass and mirrors close to the camera (nested refraction, deep recursion); 6: meshes
    # 7 (seeds >= 9000): extreme sizes side by side -- specks of 0.02 next to boulders of 20, seen from far away: what
    #    the conservative shortcuts' distance limits ("within 100 radii") and paddings are for
    style = seed % 8 if seed >= 9000 else seed % 7 if seed >= 4000 else seed % 5
    objs = []
    if rng.random() < 0.7 and style not in (1, 7):
        objs.append(P.Plane(P.translation(0.0, float(rng.uniform(-3.5, -2.0)), 0.0), _material(P, rng)))
    if style == 0:
        objs += [_leaf(P, rng) for _ in range(int(rng.integers(1, 7)))]
    elif style == 1:
        for _ in range(int(rng.integers(16, 40))):
            t = P.chain(P.translation(*[float(v) for v in rng.uniform(-3.5, 3.5, 3)]), P.scaling(*[float(v) for v in rng.uniform(0.3, 0.8, 3)]))
            objs.append((P.Sphere if rng.random() < 0.6 else P.Cube)(t, _material(P, rng)))
    elif style == 2:
        budget = [int(rng.integers(2, 7))]
        while budget[0] > 0:
            if rng.random() < 0.7:
                objs.append(_group(P, rng, 1, budget))
            else:
                budget[0] -= 1
                objs.append(_leaf(P, rng))
    elif style == 5:
        for _ in range(int(rng.integers(3, 9))):
            t = P.chain(P.translation(*[float(v) for v in rng.uniform(-1.5, 1.5, 3)]), P.rotation_z(float(rng.uniform(-1, 1))),
                        P.scaling(*[float(v) for v in rng.uniform(0.7, 2.0, 3)]))
            u = rng.random()
            m = P.Material(color=tuple(rng.uniform(0.0, 1.0, 3)), diffuse=float(rng.uniform(0.1, 0.6)), specular=0.8, shininess=120.0,
                           reflective=float(rng.uniform(0.3, 1.0)) if u < 0.5 else 0.0, transparency=float(rng.uniform(0.5, 1.0)) if u > 0.3 else 0.0,
                           refractive_index=float(rng.choice([1.0, 1.1, 1.5, 2.0])))
            objs.append([P.Sphere, P.Sphere, P.Cube, P.Cylinder][int(rng.integers(0, 4))](t, m))
    elif style == 7:
        for _ in range(int(rng.integers(6, 30))):
            r = float(10.0 ** rng.uniform(-1.7, 1.3))
            sc = np.full(3, r) * (rng.uniform(0.5, 1.5, 3) if rng.random() < 0.5 else 1.0)
            t = P.chain(P.translation(*[float(v) for v in rng.uniform(-25, 25, 3)]), P.scaling(*[float(v) for v in sc]))
            objs.append((P.Sphere if rng.random() < 0.6 else P.Cube if rng.random() < 0.7 else P.Cylinder)(t, _material(P, rng)))
    elif style == 6:
        for _ in range(int(rng.integers(1, 4))):
            objs.append(_mesh(P, rng))
        objs += [_leaf(P, rng) for _ in range(int(rng.integers(0, 3)))]
    else:
        budget = [int(rng.integers(8, 20))]
        while budget[0] > 0:
            objs.append(_group(P, rng, 0, budget))
        if rng.random() < 0.7:
            objs.append(_mesh(P, rng))
        if style == 4:
            objs += [_leaf(P, rng) for _ in range(int(rng.integers(0, 4)))]
    if rng.random() < 0.5:
        light = P.PointLight(P.point(*[float(v) for v in rng.uniform(-6, 6, 3) + np.array([0, 7, -4])]), P.color(1, 1, 1))
    else:
        jitter = ("hashed", seed) if rng.random() < 0.6 else ("constant", float(rng.choice([0.0, 0.5, 1.0, 1.5, -0.25])))
        u = rng.normal(size=3)
        v = np.cross(u, rng.normal(size=3))
        light = P.RectangleLight(P.color(1.1, 1.0, 0.9), P.point(*[float(x) for x in rng.uniform(-3, 3, 3) + np.array([0, 6, -3])]),
                                 P.vector(*[float(x) for x in u / np.linalg.norm(u) * rng.uniform(0.5, 3)]), int(rng.integers(2, 5)),
                                 P.vector(*[float(x) for x in v / np.linalg.norm(v) * rng.uniform(0.5, 3)]), int(rng.integers(2, 5)), jitter)
    w, h = int(rng.integers(40, 90)), int(rng.integers(30, 70))
    cam_z = -9.0 if style != 7 else float(-10.0 ** rng.uniform(1.0, 2.5))
    camera = (w, h, float(rng.uniform(0.6, 1.3)), P.view_transform(P.point(*[float(x) for x in rng.uniform(-2, 2, 3) + np.array([0, 1.5, cam_z])]),
                                                                   P.point(0, 0, 0), P.vector(0, 1, 0)))
    return P.World(objs, light), camera, int(rng.integers(0, 6)) if seed < 14000 else int(rng.integers(5, 9))  # up to RTC_MAX_DEPTH


# RTC_FUZZ_SEEDS=a:b widens the search (development); the default range is what the suite runs
# 2133: a cone's stray root (cone.rs:99-107) outside its group's box, which distance pruning used to skip
_SEEDS = (range(*[int(v) for v in os.environ["RTC_FUZZ_SEEDS"].split(":")]) if os.environ.get("RTC_FUZZ_SEEDS")
          else list(range(60)) + [2133] + list(range(4002, 4030, 7)) + list(range(4003, 4031, 7)) + list(range(9007, 9040, 8)) +  # + nested glass, + meshes, + extreme sizes
             [14005, 14013, 14021, 14029])  # + recursion depth 5..8 on the glass-and-mirror style


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_worlds_match_the_oracle(seed, monkeypatch):
    # the same construction script against the product's API and against the oracle's: each side bakes transforms,
    # divides and caches group boxes (stale ones included, group.rs:15) by itself
    world, cam, depth = _world(seed, P)
    own, cam_o, _ = _world(seed, O)
    camera = P.Camera(*cam)
    assert np.array_equal(np.asarray(cam[3], dtype=f32), np.asarray(cam_o[3], dtype=f32))
    exp, rays = H.oracle_camera(camera).render(own, depth, threads=8)
    names = []
    for specialise in ("0", "1"):  # ahead-of-time kernels / compiled for the scene (with all that implies: lanes per pixel, ...)
        monkeypatch.setenv("RTC_AMD_SPECIALIZE", specialise)
        r = Renderer(world, camera, device=0)
        names.append(r.kernel_name)
        # a scene's first frame, and three of the later ones, whose blocks are scheduled -- and, with several lanes per pixel,
        # re-cut -- by the frames before (rtc_device.hip refine_block_list, order_grid)
        for frame in range(4):
            img = r.render(depth).cpu().numpy()
            H.assert_images_equal(img, exp, "seed %d (%s) frame %d" % (seed, r.kernel_name, frame))
            assert r.stats()["rays"] == rays, (seed, r.kernel_name, frame)
        r.close()


# 84: a shade point 4 000 units out on a plane whose shadow ray ends beside a small sphere -- the f32 quadratic is off by two units
# there, and distance pruning used to skip the sphere's group (rtc_kernel_core.h for_each_object, the q tmin^2 term)
@pytest.mark.parametrize("seed", [3, 9, 14, 21, 24, 33, 38, 47, 84])
def test_random_worlds_at_a_size_that_takes_the_default_fast_paths(seed):
    """The same worlds at 640x420 (> 2^18 pixels) with nothing forced: whatever the library's own policies choose
    -- scene-compiled kernels, lanes per pixel by frame size, ... -- must render the oracle's image."""
    world, cam, depth = _world(seed, P)
    own, _, _ = _world(seed, O)
    camera = P.Camera(640, 420, cam[2], cam[3])
    exp, rays = H.oracle_camera(camera).render(own, depth, threads=8)
    r = Renderer(world, camera, device=0)
    for frame in range(4):  # (the first frame and frames scheduled by the frames before)
        img = r.render(depth).cpu().numpy()
        H.assert_images_equal(img, exp, "seed %d (%s) frame %d" % (seed, r.kernel_name, frame))
        assert r.stats()["rays"] == rays
    r.close()


@pytest.mark.parametrize("seed", range(0, 40, 2))
def test_random_rays_through_the_batched_entry_points(seed):
    """World::color_at for arbitrary rays (rtc_color_at: no camera, so no assumption about where rays start -- the
    library's hierarchy stays off, triangle pre-culling only serves rays that start inside its ball)."""
    world, _, _ = _world(seed, P)
    own, _, _ = _world(seed, O)
    rng = np.random.default_rng(seed)
    n = 400
    o = np.concatenate([rng.uniform(-9, 9, (n, 3)) * rng.choice([1.0, 1.0, 8.0], (n, 1)), np.ones((n, 1))], axis=1).astype(f32)
    target = rng.uniform(-3, 3, (n, 3))
    d = target - o[:, :3].astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = np.concatenate([d, np.zeros((n, 1))], axis=1).astype(f32)
    got = world.color_at(o, d, 3)
    for i in range(n):
        own.set_pixel(i)
        exp = own.color_at(o[i], d[i], 3)
        assert (got[i] == exp).all() or (np.isnan(got[i]) == np.isnan(exp)).all() and (got[i][~np.isnan(exp)] == exp[~np.isnan(exp)]).all(), (seed, i, o[i], d[i], got[i], exp)


@pytest.mark.parametrize("seed", [2, 5, 9, 13, 16, 4002, 4003])
@pytest.mark.parametrize("specialise", ["0", "1"])
def test_random_worlds_in_bands(seed, specialise, monkeypatch):
    """The multi-GPU split on random worlds: 8-row bands dealt round-robin over 3 parts, each rendered on its own and laid
    back into place, must be the whole frame -- pixel for pixel, ray for ray (the jitter key is the GLOBAL pixel index)."""
    monkeypatch.setenv("RTC_AMD_SPECIALIZE", specialise)
    world, cam, depth = _world(seed, P)
    camera = P.Camera(*cam)
    r = Renderer(world, camera, device=0)
    whole = r.render(depth).cpu().numpy()
    st = r.stats()
    h, band, n = camera.height, 8, 3
    frame = np.full_like(whole, np.nan)
    rays = 0
    for p in range(n):
        part = r.partition(band, n, p)
        rows = r.render(depth, part=part).cpu().numpy()
        rays += r.stats()["rays"]
        at = 0
        for b in range(p, (h + band - 1) // band, n):
            y0, y1 = b * band, min((b + 1) * band, h)
            frame[y0:y1] = rows[at:at + y1 - y0]
            at += y1 - y0
        assert at == rows.shape[0]
    H.assert_images_equal(frame, whole, "seed %d in bands" % seed)
    assert rays == st["rays"]
    r.close()
