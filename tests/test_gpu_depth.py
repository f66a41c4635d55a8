"""The reference's depth domain (-m gpu).  `Camera::render(world, reflection_recursion_depth: i16)` (camera.rs:76) takes
any depth -- its author renders reflect_refract at 20 -- and the recursion (world.rs:121-162) really goes that deep between
mirrors.  The kernels keep RTC_STACK_DEPTH_BASE = 8 frames per lane; above that the scene's kernel is compiled once more
with a longer frame stack (rtc_device.hip deep_kernel), and every such frame must equal the oracle's recursion bit for
bit: whole frames at the sizes the scenes are timed at, every kernel family, and -- for the jitter's path code, a u32 that
wraps beyond 31 levels in oracle and kernel alike -- an area light between facing mirrors at depth 40."""
import os

import numpy as np
import pytest

import ray_tracer_challenge_amd as P
from ray_tracer_challenge_amd import _lib as L
from ray_tracer_challenge_amd import scenes
from tests import helpers as H

pytestmark = pytest.mark.gpu
f32 = np.float32
THREADS = min(16, len(os.sched_getaffinity(0)))


def _renderer(world, camera):
    from ray_tracer_challenge_amd.renderer import Renderer
    return Renderer(world, camera, device=0)


def _check(world, camera, depth, what, rows=None):
    r = _renderer(world, camera)
    img = r.render(depth).cpu().numpy()
    st = r.stats()
    name = r.kernel_name
    r.close()
    if rows is None:
        exp, exp_rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=THREADS)
        H.assert_images_equal(img, exp, "%s depth %d (%s)" % (what, depth, name))
        assert st["rays"] == exp_rays, (what, depth, st["rays"], exp_rays)
    else:
        oc, ow = H.oracle_camera(camera), H.oracle_world(world)
        for y0, y1 in rows:
            exp, _ = oc.render(ow, depth, threads=THREADS, rows=(y0, y1))
            H.assert_images_equal(img[y0:y1], exp[y0:y1], "%s depth %d rows %d..%d" % (what, depth, y0, y1))
    return img, st


def hall_of_mirrors(width, height, light, glass_ball=True):
    """Two facing mirror walls, a mirror floor, a glass ball and a matte one between them: rays bounce until the depth runs
    out, and with glass in the way both children of a hit go deep (the ray tree grows with the depth: keep it below ~50).
    glass_ball=False: mirrors only -- one child per hit, a chain of depth + 1 rays per pixel."""
    wall = P.Material(color=(0.9, 0.9, 1.0), ambient=0.05, diffuse=0.2, specular=0.6, shininess=80.0, reflective=0.9)
    left = P.Plane()
    left.set_transformation(P.translation(-3.0, 0.0, 0.0) @ P.rotation_z(f32(-np.pi / 2)))
    left.set_material(wall)
    right = P.Plane()
    right.set_transformation(P.translation(3.0, 0.0, 0.0) @ P.rotation_z(f32(np.pi / 2)))
    right.set_material(wall.copy())
    floor = P.Plane()
    floor.set_material(P.Material(color=(0.6, 0.5, 0.4), reflective=0.5, specular=0.1))
    glass = P.Sphere()
    glass.set_transformation(P.translation(-0.7, 1.0, 0.5))
    glass.set_material(P.Material(color=(0.05, 0.05, 0.1), diffuse=0.1, specular=1.0, shininess=300.0, reflective=0.9, transparency=0.9,
                                  refractive_index=1.5))
    matte = P.Sphere()
    matte.set_transformation(P.translation(1.2, 0.6, -0.4) @ P.scaling(0.6, 0.6, 0.6))
    matte.set_material(P.Material(color=(1.0, 0.3, 0.2), reflective=0.2))
    world = P.World([left, right, floor, glass, matte] if glass_ball else [left, right, floor, matte], light)
    camera = P.Camera(width, height, f32(np.pi / 2.2), P.view_transform(P.point(0.3, 1.4, -4.5), P.point(0.0, 1.0, 0.0), P.vector(0, 1, 0)))
    return world, camera


AREA = lambda: P.RectangleLight((1.2, 1.2, 1.2), P.point(-0.5, 3.5, -0.5), P.vector(1, 0, 0), 4, P.vector(0, 0, 1), 3, jitter=("hashed", 77))  # noqa: E731
POINT = lambda: P.PointLight(P.point(0.5, 3.5, -2.0), (1.0, 1.0, 1.0))  # noqa: E731


@pytest.mark.parametrize("depth", [8, 9, 16, 17, 24, 40])
def test_hall_of_mirrors_point_light(depth):
    world, camera = hall_of_mirrors(160, 112, POINT())
    _, st = _check(world, camera, depth, "hall of mirrors, point light")
    assert st["rays"] > 160 * 112 * min(depth, 12)   # the recursion does go deep


@pytest.mark.parametrize("depth", [12, 33, 40])
def test_hall_of_mirrors_area_light_path_codes_wrap_like_the_oracles(depth):
    # 12 cells, hashed jitter: every shade point's samples are keyed by its path code (2p / 2p + 1 per level, a u32)
    world, camera = hall_of_mirrors(96, 64, AREA())
    _check(world, camera, depth, "hall of mirrors, area light")


def test_each_stack_size_against_the_oracle():
    """reflect_refract at depth 8 (the base kernels), 9 (a 16-level stack), 20 (32 levels: the author's own render): each
    frame equals the oracle at its own depth -- the glass sphere's inner reflections keep the recursion alive well beyond
    20 levels (and the ray tree growing: depth 70 takes the oracle minutes), so these are different images."""
    world, camera, _ = scenes.reflect_refract(320, 160)
    imgs = {}
    for depth in (8, 9, 20):
        imgs[depth], _ = _check(world, camera, depth, "reflect_refract 320x160")
    assert not np.array_equal(imgs[8], imgs[9]) and not np.array_equal(imgs[9], imgs[20])


@pytest.mark.parametrize("scene,size,depth,env", [
    ("hexagons", (400, 200), 12, {}),                       # GroupShapes: the packet walk
    ("sphere_grid", (512, 512), 12, {}),                    # the library's own hierarchy over a flat world
    ("sphere_grid", (384, 384), 11, {"RTC_AMD_BVH": "0"}),  # ... and the any-count loop
    ("mesh", (230, 170), 10, {}),                           # divided meshes: lanes splitting leaf runs, block lists
    ("first_textures", (300, 150), 10, {}),                 # gates + patterns + area light
    ("glass_and_mirror", (256, 256), 20, {"RTC_AMD_SPECIALIZE": "0"}),  # the policy says ahead-of-time: deep frames compile anyway
])
def test_every_kernel_family_renders_deep(scene, size, depth, env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    world, camera, _ = getattr(scenes, scene)(*size)
    _check(world, camera, depth, scene)


def test_reflect_refract_at_the_authors_depth_whole_frame():
    """reflect_refract.rs at depth 20 (the reference's todo.md), at the size the scene is timed at: every pixel, the ray count."""
    world, camera, _ = scenes.reflect_refract(4096, 2048)
    _check(world, camera, 20, "reflect_refract 4096x2048")


def test_c4_at_depth_20_whole_frame():
    world, camera, _ = scenes.glass_and_mirror(4096, 4096)
    _check(world, camera, 20, "C4 glass_and_mirror 4096^2")


def test_the_depth_domain_ends_where_the_header_says():
    world, camera = hall_of_mirrors(40, 24, POINT(), glass_ball=False)  # (a chain per pixel: 256 rays, not 2^256)
    r = _renderer(world, camera)
    with pytest.raises(L.RtcError) as e:
        r.render(L.RTC_MAX_DEPTH + 1)
    assert e.value.status == L.RTC_ERR_INVALID_ARG
    with pytest.raises(L.RtcError):
        r.render(-1)
    r.close()
    # ... and the last depth inside it renders (a 255-level stack), through the one-call seam as well
    img, _ = _check(world, camera, L.RTC_MAX_DEPTH, "hall of mirrors at RTC_MAX_DEPTH")
    canvas = camera.render(world, L.RTC_MAX_DEPTH)
    assert np.array_equal(canvas.data.view(np.uint32), img.view(np.uint32))
    # the batched test entry point keeps the base stack
    o = np.array([[0.3, 1.4, -4.5, 1.0]], dtype=f32)
    d = np.array([[0.0, 0.0, 1.0, 0.0]], dtype=f32)
    with pytest.raises(L.RtcError):
        world.color_at(o, d, L.RTC_STACK_DEPTH_BASE + 1)
