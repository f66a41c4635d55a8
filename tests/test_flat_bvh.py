"""The library's own bounding-volume hierarchy over FLAT worlds of many bounded objects (rtc_device.hip
build_flat_bvh): unlike a GroupShape's box it is not part of the reference's semantics (world.rs:60-77 tests
every object, in order), so it must never change a bit of the image, a ray count or a shaded-hit count.
Checked against the oracle, which knows nothing of it, and against the same kernel family with it switched off."""
import numpy as np
import pytest

import ray_tracer_challenge_amd as P
from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.renderer import Renderer
from tests import helpers as H

pytestmark = pytest.mark.gpu
f32 = np.float32


def _cloud(seed, n, duplicates=0, glass=0.3, mirror=0.3, cubes=0.4, rect_light=False):
    """n overlapping spheres and cubes (scale + translate only) in a 10-unit cloud.  `duplicates` objects are
    re-listed with the SAME transform and another material: every hit on them is an exact tie in t."""
    rng = np.random.default_rng(seed)
    objs = []
    for _ in range(n):
        r = rng.uniform(0.3, 1.2, 3) if rng.random() < 0.5 else np.full(3, rng.uniform(0.3, 1.2))
        c = rng.uniform(-5.0, 5.0, 3)
        u = rng.random()
        m = P.Material(color=tuple(rng.uniform(0.1, 1.0, 3)), diffuse=0.7, specular=0.4,
                       reflective=0.5 if u < mirror else 0.0,
                       transparency=0.8 if mirror <= u < mirror + glass else 0.0,
                       refractive_index=float(rng.choice([1.0, 1.3, 1.5, 2.0])))
        t = P.chain(P.translation(*map(float, c)), P.scaling(*map(float, r)))
        objs.append((P.Cube if rng.random() < cubes else P.Sphere)(t, m))
    for k in range(duplicates):
        src = objs[int(rng.integers(0, n))]
        m = P.Material(color=tuple(rng.uniform(0.1, 1.0, 3)), transparency=0.6 if k % 2 else 0.0,
                       refractive_index=1.7, reflective=0.2)
        objs.insert(int(rng.integers(0, len(objs) + 1)), P.Shape(src.kind, src.transform, m))
    if rect_light:
        light = P.RectangleLight(P.color(1, 1, 1), P.point(-3, 9, -9), P.vector(2, 0, 0), 4, P.vector(0, 2, 0), 4,
                                 jitter=("hashed", 99 + seed))
    else:
        light = P.PointLight(P.point(-8, 12, -10), P.color(1, 1, 1))
    cam = P.Camera(96, 80, float(np.pi / 3), P.view_transform(P.point(1, 4, -14), P.point(0, 0, 0), P.vector(0, 1, 0)))
    return P.World(objs, light), cam


def _render(world, camera, depth, bvh, monkeypatch, expect=None):
    monkeypatch.setenv("RTC_AMD_BVH", "1" if bvh else "0")
    r = Renderer(world, camera, device=0)
    if expect is not None:
        assert ("tree,bvh" in r.kernel_name) == expect, r.kernel_name
    img = r.render(depth).cpu().numpy()
    st = r.stats()
    r.close()
    return img, st


@pytest.mark.parametrize("seed,n,dups,rect", [(1, 16, 0, False), (2, 23, 4, False), (3, 40, 8, False), (4, 64, 0, True),
                                             (5, 31, 6, True), (6, 100, 10, False)])
@pytest.mark.parametrize("specialise", ["0", "1"])  # ahead-of-time traversal kernel / compiled for the scene (hiprtc)
def test_flat_bvh_matches_oracle_and_the_flat_loop(seed, n, dups, rect, specialise, monkeypatch):
    world, camera = _cloud(seed, n, duplicates=dups, rect_light=rect)
    monkeypatch.setenv("RTC_AMD_SPECIALIZE", specialise)
    depth = 4
    on, st_on = _render(world, camera, depth, True, monkeypatch, expect=True)
    off, st_off = _render(world, camera, depth, False, monkeypatch, expect=False)
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=8)
    H.assert_images_equal(on, exp, "bvh on, seed %d" % seed)
    H.assert_images_equal(off, exp, "bvh off, seed %d" % seed)
    assert st_on["rays"] == rays == st_off["rays"]
    assert st_on["shaded_hits"] == st_off["shaded_hits"]


def test_coincident_objects_resolve_ties_by_list_order(monkeypatch):
    """Three identical spheres with different materials, buried among others so that the median split separates
    them: the reference's sort is stable, the first listed wins the nearest hit (world.rs:60-77 + intersection.rs
    hit()) and the n1/n2 walk removes/appends containers in list order (intersection.rs:95-133)."""
    world, camera = _cloud(11, 24, glass=0.6, mirror=0.2, cubes=0.0)
    t = P.chain(P.translation(0.0, 0.0, -2.0), P.scaling(2.0, 2.0, 2.0))
    for k, (idx, col) in enumerate([(2, (1, 0, 0)), (13, (0, 1, 0)), (27, (0, 0, 1))]):
        world.objects.insert(idx, P.Sphere(t, P.Material(color=col, transparency=0.9 if k != 1 else 0.0,
                                                        refractive_index=1.0 + 0.25 * k, reflective=0.1 * k)))
    on, st_on = _render(world, camera, 5, True, monkeypatch, expect=True)
    off, st_off = _render(world, camera, 5, False, monkeypatch, expect=False)
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), 5, threads=8)
    H.assert_images_equal(on, exp, "coincident spheres, bvh on")
    H.assert_images_equal(off, exp, "coincident spheres, bvh off")
    assert st_on["rays"] == rays == st_off["rays"]


def test_flat_bvh_eligibility(monkeypatch):
    monkeypatch.setenv("RTC_AMD_BVH", "1")
    monkeypatch.setenv("RTC_AMD_SPECIALIZE", "0")
    world, camera = _cloud(7, 20)
    assert Renderer(world, camera, device=0).kernel_name == "render_kernel<tree,bvh>"
    monkeypatch.setenv("RTC_AMD_SPECIALIZE", "1")
    assert Renderer(world, camera, device=0).kernel_name == "render_kernel_spec[tree,bvh]"      # spheres and cubes
    balls = P.World([o for o in world.objects if o.kind == world.objects[0].kind] * 2, world.light)
    assert Renderer(balls, camera, device=0).kernel_name.startswith("render_kernel_spec[tree,bvh;all 0x")
    monkeypatch.setenv("RTC_AMD_SPECIALIZE", "0")
    few = P.World(world.objects[:15], world.light)                    # too few objects to pay
    assert Renderer(few, camera, device=0).kernel_name.find("bvh") < 0
    floor = P.World(world.objects + [P.Plane()], world.light)         # unbounded object: ray origins unbounded
    assert Renderer(floor, camera, device=0).kernel_name.find("bvh") < 0
    tilted = P.World(list(world.objects), world.light)
    tilted.objects[3] = P.Sphere(P.chain(P.rotation_z(0.3), P.scaling(1, 2, 1)), world.objects[3].material)
    assert Renderer(tilted, camera, device=0).kernel_name.find("bvh") < 0
    speck = P.World(list(world.objects), world.light)                 # a 0.01-radius sphere: > 100 radii from the camera
    speck.objects[5] = P.Sphere(P.scaling(0.01, 0.01, 0.01), world.objects[5].material)
    assert Renderer(speck, camera, device=0).kernel_name.find("bvh") < 0
    cyl = P.World(list(world.objects), world.light)
    cyl.objects[0] = P.Cylinder(P.identity_4x4(), world.objects[0].material, minimum_y=0.0, maximum_y=1.0)
    assert Renderer(cyl, camera, device=0).kernel_name.find("bvh") < 0


@pytest.mark.parametrize("size", [(1024, 1024), (2048, 1536)])
def test_flat_bvh_equals_flat_loop_at_larger_sizes(size, monkeypatch):
    """sphere_grid (BASELINE C5's scene) and a dense cloud at sizes the oracle does not reach in seconds:
    hierarchy on == hierarchy off, bit for bit, with equal ray and shaded-hit counts."""
    for world, camera, depth in (scenes.sphere_grid(*size), _cloud(21, 80, duplicates=6) + (5,)):
        if not hasattr(camera, "width") or camera.width != size[0]:
            camera = P.Camera(size[0], size[1], camera.field_of_view, camera.transform)
        on, st_on = _render(world, camera, depth, True, monkeypatch, expect=True)
        off, st_off = _render(world, camera, depth, False, monkeypatch, expect=False)
        H.assert_images_equal(on, off, "hierarchy on vs off")
        assert st_on["rays"] == st_off["rays"] and st_on["shaded_hits"] == st_off["shaded_hits"]
