"""Pins the CPU oracle to the reference's own known-answer tests for the
render hot path (shapes, intersection, world, lights, Phong, canvas).
Vectors: tests/golden/reference_kat.json (transcribed from the reference's
inline #[test] functions; each block cites file:line).  No GPU needed.
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests import kat as K

f32 = np.float32
S2 = K.CONSTS["FRAC_1_SQRT_2"]
SQRT_2 = K.CONSTS["SQRT_2"]


# ------------------------------------------------------------------ shapes
def test_sphere_intersections(kat):  # ray.rs:75-120
    s = O.Sphere()
    for o, d, expected in kat["sphere"]["intersect"]["cases"]:
        ts, _, _ = s.intersect(K.point(o), K.vector(d))
        K.assert_exact(ts, expected)
    c = kat["sphere"]["scaled"]
    s2 = O.Sphere(O.scaling(2.0, 2.0, 2.0))
    K.assert_exact(s2.local_intersect(K.point(c["ray"][0]), K.vector(c["ray"][1])), c["expect_exact"])
    c = kat["sphere"]["translated_miss"]
    s3 = O.Sphere(O.translation(5.0, 0.0, 0.0))
    assert s3.local_intersect(K.point(c["ray"][0]), K.vector(c["ray"][1])) == []


def test_sphere_normals(kat):  # shape/sphere.rs:114-145
    s = O.Sphere()
    for p, n in kat["sphere"]["normals"]["cases"]:
        K.assert_exact(s.local_norm_at(K.point(p)), K.vec(n))
    q = f32(1.0) / np.sqrt(f32(3.0))
    K.assert_eps(s.local_norm_at(K.point([q, q, q])), np.array([q, q, q, 0], dtype=f32))


def test_plane(kat):  # shape/plane.rs:74-116
    p = O.Plane()
    for pt in kat["plane"]["normal_points"]:
        K.assert_exact(p.local_norm_at(K.point(pt)), [0, 1, 0, 0])
    for o, d, expected in kat["plane"]["intersect"]:
        K.assert_exact(p.local_intersect(K.point(o), K.vector(d)), expected)


def test_cube(kat):  # shape/cube.rs:136-230
    c = O.Cube()
    for o, d, t0, t1 in kat["cube"]["hits"]["cases"]:
        K.assert_exact(c.local_intersect(K.point(o), K.vector(d)), [t0, t1])
    for o, d in kat["cube"]["misses"]["cases"]:
        assert c.local_intersect(K.point(o), K.vector(d)) == []
    for p, n in kat["cube"]["normals"]["cases"]:
        K.assert_exact(c.local_norm_at(K.point(p)), K.vector(n))


def test_aabb_intersection(kat):  # bounding_box.rs:199-247
    for key in ("at_origin", "off_origin"):
        blk = kat["aabb"][key]
        for o, d, expected in blk["cases"]:
            got = O.aabb_intersection(K.point(o), O.norm(K.vector(d)), K.point(blk["min"]), K.point(blk["max"]))
            assert (got is not None) == expected, (key, o, d)


def test_cylinder(kat):  # shape/cylinder.rs:161-363
    cy = kat["cylinder"]
    c = O.Cylinder()
    for o, d in cy["misses"]["cases"]:
        assert c.local_intersect(K.point(o), O.norm(K.vector(d))) == []
    for o, d, t0, t1 in cy["sides"]["cases"]:
        ts = c.local_intersect(K.point(o), O.norm(K.vector(d)))
        assert len(ts) == 2
        K.assert_eps(ts, [t0, t1])
    for key in ("constrained", "caps"):
        blk = cy[key]
        cc = O.Cylinder(minimum_y=blk["min"], maximum_y=blk["max"], closed=blk["closed"])
        for o, d, count in blk["cases"]:
            assert len(cc.local_intersect(K.point(o), O.norm(K.vector(d)))) == count, (key, o, d)
    for p, n in cy["side_normals"]["cases"]:
        K.assert_exact(c.local_norm_at(K.point(p)), K.vector(n))
    blk = cy["cap_normals"]
    cc = O.Cylinder(minimum_y=blk["min"], maximum_y=blk["max"], closed=blk["closed"])
    for p, n in blk["cases"]:
        K.assert_exact(cc.local_norm_at(K.point(p)), K.vector(n))


def test_shape_object_space_ray_and_normals(kat):  # shape/shape.rs:203-243
    sh = kat["shape"]
    c = sh["scaled_ray"]
    _, oo, od = O.Sphere(O.scaling(*c["scaling"])).intersect(K.point(c["ray"][0]), K.vector(c["ray"][1]))
    K.assert_exact(oo, c["object_ray"][0])
    K.assert_exact(od, c["object_ray"][1])
    c = sh["translated_ray"]
    _, oo, od = O.Sphere(O.translation(*c["translation"])).intersect(K.point(c["ray"][0]), K.vector(c["ray"][1]))
    K.assert_exact(oo, c["object_ray"][0])
    K.assert_exact(od, c["object_ray"][1])
    c = sh["normal_translated"]
    n = O.TestShape(O.translation(*c["translation"])).normal_at(K.point(c["point"]))
    K.assert_eps(n, c["expect_eps"])
    c = sh["normal_transformed"]
    t = O.mat_mul(O.scaling(1.0, 0.5, 1.0), O.rotation_z(K.CONSTS["PI"] / f32(5.0)))
    n = O.TestShape(t).normal_at(K.point(c["point"]))
    K.assert_eps(n, c["expect_eps"])
    n = O.TestShape().normal_at(K.point([1, 5, 10]))  # normal_is_normalized_vector
    K.assert_eps(n, O.norm(n))


def test_hit_selection(kat):  # intersection.rs:53-95
    for ts, expected in kat["intersection_hit"]["cases"]:
        assert O.hit(ts) == expected
    assert O.hit([f32(-0.0), 1.0]) == 0          # -0.0 >= 0.0 is true
    assert O.hit([2.0, 1.0, 1.0]) == 1           # first of equal minima


def test_ray_position(kat):  # ray.rs:67-73
    c = kat["ray"]["position"]
    for t, expected in c["cases"]:
        K.assert_exact(O.position(K.point(c["ray"][0]), K.vector(c["ray"][1]), t), expected)


# ------------------------------------------------------------------- world
def test_intersect_world_with_ray(kat):  # world.rs:322-332
    c = kat["world"]["intersect_world_with_ray"]
    ts, objs = O.default_world().intersect(K.point(c["ray"][0]), K.vector(c["ray"][1]))
    K.assert_exact(ts, c["expect_exact"])
    assert list(objs) == [0, 1, 1, 0]


def test_precompute(kat):  # world.rs:334-381
    w = O.World([O.Sphere()], O.PointLight(O.point(0, 0, 0), O.color(1, 1, 1)))
    for key in ("precompute_state", "precompute_inside"):
        c = kat["world"][key]
        comps = w.precompute_values(K.point(c["ray"][0]), K.vector(c["ray"][1]), 0, [(c["t"], 0)])
        K.assert_exact(O.arr4(comps.point), c["point"])
        K.assert_exact(O.arr4(comps.eye), c["eye"])
        K.assert_exact(O.arr4(comps.normal), c["normal"])
        assert bool(comps.inside) == c["inside"]
        assert comps.distance == c["t"]
    c = kat["world"]["precompute_reflection_vector"]
    w = O.World([O.Plane()], O.PointLight(O.point(0, 0, 0), O.color(1, 1, 1)))
    comps = w.precompute_values(K.point(c["ray"][0]), K.vector(c["ray"][1]), 0, [(K.val(c["t"]), 0)])
    K.assert_exact(O.arr4(comps.reflectv), K.vec(c["reflectv"]))


def _glass(transform, ri):
    return O.Sphere(transform, O.Material(transparency=1.0, refractive_index=ri))


def test_find_n1_and_n2(kat):  # world.rs:396-451
    c = kat["world"]["find_n1_and_n2"]
    w = O.World([_glass(O.scaling(2.0, 2.0, 2.0), 1.5), _glass(O.translation(0.0, 0.0, -0.25), 2.0),
                 _glass(O.translation(0.0, 0.0, 0.25), 2.5)], O.PointLight(O.point(0, 0, 0), O.color(1, 1, 1)))
    xs = [tuple(x) for x in c["xs"]]
    for i, (n1, n2) in enumerate(c["expect_exact"]):
        comps = w.precompute_values(K.point(c["ray"][0]), K.vector(c["ray"][1]), i, xs)
        assert comps.n1 == n1 and comps.n2 == n2, (i, comps.n1, comps.n2)


def test_over_and_under_point_offsets():  # world.rs:453-466, 633-643
    eps = f32(1.1920929e-7) * f32(10000.0)
    w = O.World([_glass(O.translation(0.0, 0.0, 1.0), 1.5)], O.PointLight(O.point(0, 0, 0), O.color(1, 1, 1)))
    comps = w.precompute_values(O.point(0, 0, -5), O.vector(0, 0, 1), 0, [(5.0, 0)])
    assert comps.under_point[2] > eps / f32(2.0)
    assert comps.point[2] < comps.under_point[2]
    assert comps.over_point[2] < -eps / f32(2.0)
    assert comps.over_point[2] > -eps * f32(2.0)
    assert comps.point[2] > comps.over_point[2]


def _reflective_plane_world():
    w = O.default_world()
    w.objects.append(O.Plane(O.translation(0.0, -1.0, 0.0), O.Material(reflective=0.5)))
    return w


def test_reflection(kat):  # world.rs:469-537
    W = kat["world"]
    w = O.default_world()
    w.objects[1].material = w.objects[1].material.copy(ambient=1.0)
    c = W["reflected_color_nonreflective"]
    comps = w.precompute_values(K.point(c["ray"][0]), K.vector(c["ray"][1]), 0, [tuple(c["hit"])])
    K.assert_exact(w.reflected_color(comps, c["depth"]), c["expect_exact"])

    w = _reflective_plane_world()
    o, d = O.point(0, 0, -3), np.array([0, -S2, S2, 0], dtype=f32)
    comps = w.precompute_values(o, d, 0, [(SQRT_2, 2)])
    K.assert_eps(w.reflected_color(comps, 1), W["reflected_color_reflective"]["expect_eps"])
    K.assert_eps(w.shade_hit(comps, 1), W["shade_hit_reflective"]["expect_eps"])
    K.assert_eps(w.reflected_color(comps, 0), W["reflected_color_at_max_depth"]["expect_eps"])


def test_mutually_reflective_surfaces_terminate():  # world.rs:511-523
    m = O.Material(reflective=1.0)
    w = O.World([O.Plane(O.translation(0.0, -1.0, 0.0), m), O.Plane(O.translation(0.0, 1.0, 0.0), m)],
                O.PointLight(O.point(0, 0, 0), O.color(0, 0, 0)))
    w.color_at(O.point(0, 0, 0), O.vector(0, 1, 0), 1)


def test_shading(kat):  # world.rs:540-590, 646-658
    W = kat["world"]
    w = O.default_world()
    c = W["shade_intersection"]
    comps = w.precompute_values(K.point(c["ray"][0]), K.vector(c["ray"][1]), 0, [tuple(c["hit"])])
    K.assert_eps(w.shade_hit(comps, c["depth"]), c["expect_eps"])

    c = W["shade_intersection_from_inside"]
    w = O.default_world()
    w.light = O.PointLight(K.point(c["light"]), O.color(1, 1, 1))
    comps = w.precompute_values(K.point(c["ray"][0]), K.vector(c["ray"][1]), 0, [tuple(c["hit"])])
    K.assert_eps(w.shade_hit(comps, c["depth"]), c["expect_eps"])

    w = O.default_world()
    c = W["color_when_ray_misses"]
    K.assert_exact(w.color_at(K.point(c["ray"][0]), K.vector(c["ray"][1]), c["depth"]), c["expect_exact"])
    c = W["color_when_ray_hits"]
    K.assert_eps(w.color_at(K.point(c["ray"][0]), K.vector(c["ray"][1]), c["depth"]), c["expect_eps"])

    c = W["color_when_intersection_behind_ray"]
    w = O.default_world()
    w.objects[0].material = O.Material(ambient=1.0)
    w.objects[1].material = O.Material(ambient=1.0)
    K.assert_exact(w.color_at(K.point(c["ray"][0]), K.vector(c["ray"][1]), c["depth"]), c["expect_exact"])

    c = W["shade_hit_in_shadow"]
    w = O.World([O.Sphere(), O.Sphere(O.translation(*c["s2_translation"]))],
                O.PointLight(K.point(c["light"]), O.color(1, 1, 1)))
    comps = w.precompute_values(K.point(c["ray"][0]), K.vector(c["ray"][1]), 0, [tuple(c["hit"])])
    K.assert_exact(w.shade_hit(comps, c["depth"]), K.vec(c["expect_exact"]))


def test_shadows_and_point_light_intensity(kat):  # world.rs:593-630
    w = O.default_world()
    c = kat["world"]["is_shadowed"]
    for p, expected in c["cases"]:
        assert w.is_shadowed(K.point(c["light_position"]), K.point(p)) == expected, p
    for p, expected in kat["world"]["point_light_intensity_at"]["cases"]:
        K.assert_eps(w.intensity_at(K.point(p)), expected)


def _transparent_sphere_world():
    w = O.default_world()
    w.objects[0].material = w.objects[0].material.copy(transparency=1.0, refractive_index=1.5)
    return w


def test_refraction(kat):  # world.rs:661-713
    W = kat["world"]
    c = W["refracted_color_opaque"]
    w = O.default_world()
    comps = w.precompute_values(K.point(c["ray"][0]), K.vector(c["ray"][1]), c["hit_index"], [tuple(x) for x in c["xs"]])
    K.assert_eps(w.refracted_color(comps, c["depth"]), c["expect_eps"])
    w = _transparent_sphere_world()
    comps = w.precompute_values(K.point(c["ray"][0]), K.vector(c["ray"][1]), c["hit_index"], [tuple(x) for x in c["xs"]])
    K.assert_eps(w.refracted_color(comps, 0), W["refracted_color_max_depth"]["expect_eps"])
    c = W["refracted_color_tir"]
    xs = [(K.val(t), o) for t, o in c["xs"]]
    comps = w.precompute_values(K.point(c["ray"][0]), K.vector(c["ray"][1]), c["hit_index"], xs)
    K.assert_eps(w.refracted_color(comps, c["depth"]), c["expect_eps"])


def _floor_and_ball_world(floor_material):
    w = O.default_world()
    w.objects.append(O.Plane(O.translation(0.0, -1.0, 0.0), floor_material))
    w.objects.append(O.Sphere(O.translation(0.0, -3.5, -0.5), O.Material(color=(1, 0, 0), ambient=0.5)))
    return w


def test_transparent_and_schlick_shading(kat):  # world.rs:747-844
    W = kat["world"]
    o, d = O.point(0, 0, -3), np.array([0, -S2, S2, 0], dtype=f32)
    w = _floor_and_ball_world(O.Material(transparency=0.5, refractive_index=1.5))
    comps = w.precompute_values(o, d, 0, [(SQRT_2, 2)])
    K.assert_eps(w.shade_hit(comps, 5), W["shade_hit_transparent"]["expect_eps"])
    w = _floor_and_ball_world(O.Material(reflective=0.5, transparency=0.5, refractive_index=1.5))
    comps = w.precompute_values(o, d, 0, [(SQRT_2, 2)])
    K.assert_eps(w.shade_hit(comps, 5), W["shade_hit_reflective_transparent"]["expect_eps"])


def test_schlick(kat):  # world.rs:780-812
    W = kat["world"]
    w = O.World([_glass(O.identity_4x4(), 1.5)], O.PointLight(O.point(0, 0, 0), O.color(1, 1, 1)))
    comps = w.precompute_values(np.array([0, 0, S2, 1], dtype=f32), O.vector(0, 1, 0), 1, [(-S2, 0), (S2, 0)])
    assert O.schlick_reflectance(comps) == W["schlick_tir"]["expect_exact"]
    comps = w.precompute_values(O.point(0, 0, 0), O.vector(0, 1, 0), 1, [(-1.0, 0), (1.0, 0)])
    K.assert_eps(O.schlick_reflectance(comps), W["schlick_perpendicular"]["expect_eps"])
    c = W["schlick_small_angle"]
    comps = w.precompute_values(K.point(c["ray"][0]), K.vector(c["ray"][1]), 0, [tuple(x) for x in c["xs"]])
    K.assert_eps(O.schlick_reflectance(comps), c["expect_eps"])


# ------------------------------------------------------------------ lights
def test_rectangle_light(kat):  # light/rectangle_light.rs:99-166
    R = kat["rectangle_light"]
    c = R["construction"]
    w = O.World([], O.RectangleLight(O.color(1, 1, 1), K.point(c["corner"]), K.vector(c["u"]), c["u_steps"],
                                     K.vector(c["v"]), c["v_steps"], ("constant", 0.5)))
    pos, u, v, cells = w.light_info()
    K.assert_exact(u, c["u_vec"])
    K.assert_exact(v, c["v_vec"])
    K.assert_exact(pos, c["position"])
    assert cells == c["cells"]

    p = R["point_on_light"]
    for u_i, v_i, expected in p["cases"]:
        w = O.World([], O.RectangleLight(O.color(1, 1, 1), K.point(c["corner"]), K.vector(c["u"]), 4,
                                         K.vector(c["v"]), 2, ("cycle", p["jitter_cycle"])))
        K.assert_exact(w.point_on_light(u_i, v_i), expected)

    c = R["intensity_at"]
    for pt, expected in c["cases"]:
        w = O.default_world()
        w.light = O.RectangleLight(O.color(1, 1, 1), K.point(c["corner"]), K.vector(c["u"]), c["steps"],
                                   K.vector(c["v"]), c["steps"], ("cycle", c["jitter_cycle"]))
        K.assert_exact(w.intensity_at(K.point(pt)), expected)


def test_phong_lighting(kat):  # light/phong_lighting.rs:78-194, 238-271
    P = kat["phong"]
    m = O.Material()
    for c in P["cases"]:
        w = O.World([], O.PointLight(K.point(c["light"]), O.color(1, 1, 1)))
        got = w.phong_lighting(m, O.point(0, 0, 0), K.vector(c["eye"]), K.vector(c["normal"]), c["intensity"])
        if "expect_exact" in c:
            e = c["expect_exact"]
            e = [K.val(e)] * 3 if isinstance(e, str) else e
            K.assert_exact(got, K.vec(e))
        else:
            K.assert_eps(got, c["expect_eps"])
    a = P["attenuation"]
    w = O.World([], O.PointLight(K.point(a["light"]), O.color(1, 1, 1)))
    m = O.Material(ambient=0.1, diffuse=0.9, specular=0.0, color=(1, 1, 1))
    for intensity, expected in a["cases"]:
        K.assert_eps(w.phong_lighting(m, K.point(a["point"]), K.vector(a["eye"]), K.vector(a["normal"]), intensity),
                     expected)


def test_hashed_jitter_is_in_open_closed_unit_interval():
    vals = [O.jitter_value(O.jitter_hash(0x5EED5EED, p, 1, c, d)) for p in range(50) for c in range(20) for d in (0, 1)]
    assert min(vals) > 0.0 and max(vals) <= 1.0
    assert O.jitter_value(0xFFFFFFFF) == f32(1.0) and O.jitter_value(0xFFFFFE00) == f32(1.0)
    assert O.jitter_value(0) == f32(2.0 ** -23) and O.jitter_value(0x1FF) == f32(2.0 ** -23)
    assert O.jitter_value(0x200) == f32(2.0 ** -22)
    # the two draws of a cell are separate hashes: (cell, 1) is keyed like no other (cell', 0)
    h = {(c, d): O.jitter_hash(0x5EED5EED, 7, 1, c, d) for c in range(100) for d in (0, 1)}
    assert len(set(h.values())) == 200
    assert abs(float(np.mean(vals)) - 0.5) < 0.02


# ------------------------------------------------------------------ camera
def test_render_world(kat):  # camera.rs:156-167
    c = kat["camera"]["render_world"]
    cam = O.Camera(*c["size"], K.CONSTS["PI"] / f32(2.0),
                   O.view_transform(K.point(c["from"]), K.point(c["to"]), K.vector(c["up"])))
    img, rays = cam.render(O.default_world(), c["depth"])
    K.assert_eps(img[c["pixel"][1], c["pixel"][0]], c["expect_eps"])
    # camera.rs:80-81: the last row and column are never traced
    assert np.all(img[-1, :, :] == 0) and np.all(img[:, -1, :] == 0)
    assert rays >= 100
    img4, rays4 = cam.render(O.default_world(), c["depth"], threads=4)
    assert np.array_equal(img, img4) and rays == rays4


# ------------------------------------------------------------------ canvas
def test_scale_color_and_ppm(kat):  # canvas.rs:218-278
    C = kat["canvas"]
    for v, expected in C["scale_color"]["cases"]:
        assert O.scale_color(v) == expected
    c = C["pixel_data"]
    w, h = c["size"]
    img = np.zeros((h, w, 3), dtype=f32)
    for x, y, col in c["pixels"]:
        img[y, x] = col
    assert O.to_ppm(img).decode().split("\n")[:-1] == c["lines"]
    c = C["long_lines"]
    w, h = c["size"]
    img = np.zeros((h, w, 3), dtype=f32)
    img[:, :] = np.array(c["fill"], dtype=f32)
    text = O.to_ppm(img).decode()
    assert text.endswith("\n")
    assert text.split("\n")[:-1] == c["lines"]
    c = C["header"]
    w, h = c["size"]
    assert O.to_ppm(np.zeros((h, w, 3), dtype=f32)).decode().split("\n")[:3] == c["lines"]


def test_cone(kat):  # shape/cone.rs:190-278
    co = kat["cone"]
    c = O.Cone()
    for o, d, t0, t1 in co["sides"]["cases"]:
        ts = c.local_intersect(K.point(o), O.norm(K.vector(d)))
        assert len(ts) == 2, (o, d, ts)
        K.assert_eps(ts, [t0, t1])
    blk = co["parallel_to_one_half"]
    ts = c.local_intersect(K.point(blk["ray"][0]), O.norm(K.vector(blk["ray"][1])))
    assert len(ts) == 1
    K.assert_eps(ts, blk["expect_eps"])
    blk = co["caps"]
    cc = O.Cone(minimum_y=blk["min"], maximum_y=blk["max"], closed=blk["closed"])
    for o, d, count in blk["cases"]:
        assert len(cc.local_intersect(K.point(o), O.norm(K.vector(d)))) == count, (o, d)
    for p, n in co["normals"]["cases"]:
        K.assert_exact(c.local_norm_at(K.point(p)), K.vector(n))


PATTERN_CTORS = {"stripes": O.Stripes, "gradient": O.Gradient, "rings": O.Rings, "checkers": O.Checkers,
                 "sine_2d": O.Sine2D}


def test_patterns_color_at_world(kat):  # pattern/{stripes,gradient,rings,checkers,sine_2d}.rs tests
    for name, ctor in PATTERN_CTORS.items():
        blk = kat["pattern"][name]
        pat = ctor(blk["a"], blk["b"])
        for p, expect in blk["cases_exact"]:
            K.assert_exact(pat.color_at_world(K.point(p)), K.vec(expect))
        for p, expect in blk.get("cases_eps", []):
            K.assert_eps(pat.color_at_world(K.point(p)), K.vec(expect))


def test_pattern_transformations(kat):  # pattern/pattern.rs:99-122
    P = kat["pattern"]
    c = P["with_object_transformation"]
    got = O.TestPattern().color_at_object(K.point(c["point"]), O.Sphere(O.scaling(*c["object_scaling"])))
    K.assert_exact(got, c["expect_exact"])
    c = P["with_pattern_transformation"]
    got = O.TestPattern(O.scaling(*c["pattern_scaling"])).color_at_object(K.point(c["point"]), O.Sphere())
    K.assert_exact(got, c["expect_exact"])
    c = P["with_both_transformations"]
    got = O.TestPattern(O.translation(*c["pattern_translation"])).color_at_object(
        K.point(c["point"]), O.Sphere(O.scaling(*c["object_scaling"])))
    K.assert_exact(got, c["expect_exact"])


def test_phong_lighting_with_pattern(kat):  # light/phong_lighting.rs:197-235
    c = kat["pattern"]["phong_with_pattern"]
    m = O.Material(color=c["color"], ambient=c["ambient"], diffuse=c["diffuse"], specular=c["specular"],
                   pattern=O.Stripes((1, 1, 1), (0, 0, 0)))
    w = O.World([], O.PointLight(K.point(c["light"]), O.color(1, 1, 1)))
    for p, expect in c["cases_exact"]:
        got = w.phong_lighting(m, K.point(p), K.vector(c["eye"]), K.vector(c["normal"]), 1.0)
        K.assert_exact(got, expect)


def test_refracted_color_with_refracted_ray(kat):  # world.rs:716-744
    c = kat["pattern"]["refracted_color_with_refracted_ray"]
    w = O.default_world()
    w.objects[0].material = w.objects[0].material.copy(ambient=1.0, pattern=O.TestPattern())
    w.objects[1].material = w.objects[1].material.copy(transparency=1.0, refractive_index=1.5)
    comps = w.precompute_values(K.point(c["ray"][0]), K.vector(c["ray"][1]), c["hit_index"], [tuple(x) for x in c["xs"]])
    K.assert_eps(w.refracted_color(comps, c["depth"]), c["expect_eps"])


# ----------------------------------------------------------- bounding_box.rs / shape/group.rs
def _box(pair):
    return O.BoundingBox(K.point(pair[0]), K.point(pair[1]))


def test_bounding_box(kat):  # bounding_box.rs:136-288
    B = kat["bounding_box"]
    c = B["add_points"]
    b = O.BoundingBox.empty()
    for p in c["points"]:
        b.add_point(K.point(p))
    K.assert_exact(b.min, K.point(c["min"]))
    K.assert_exact(b.max, K.point(c["max"]))
    c = B["add_box"]
    b = _box(c["box1"])
    b.add_bounding_box(_box(c["box2"]))
    K.assert_exact(b.min, K.point(c["min"]))
    K.assert_exact(b.max, K.point(c["max"]))
    c = B["contains_point"]
    for p, expect in c["cases"]:
        assert _box(c["box"]).contains_point(K.point(p)) == expect, p
    c = B["contains_box"]
    for mn, mx, expect in c["cases"]:
        assert _box(c["box"]).contains_bounding_box(_box([mn, mx])) == expect, (mn, mx)
    c = B["transform"]
    b2 = _box(c["box"]).transform(O.mat_mul(O.rotation_x(K.CONSTS["PI"] / f32(4.0)), O.rotation_y(K.CONSTS["PI"] / f32(4.0))))
    K.assert_eps(b2.min[:3], c["min_eps"])
    K.assert_eps(b2.max[:3], c["max_eps"])
    for c in B["split"]["cases"]:
        left, right = _box(c["box"]).split()
        K.assert_exact(left.min, K.point(c["left"][0]))
        K.assert_exact(left.max, K.point(c["left"][1]))
        K.assert_exact(right.min, K.point(c["right"][0]))
        K.assert_exact(right.max, K.point(c["right"][1]))
    # the same ray tables as cube's aabb_intersection (bounding_box.rs:199-247)
    for key in ("at_origin", "off_origin"):
        blk = kat["aabb"][key]
        b = O.BoundingBox(K.point(blk["min"]), K.point(blk["max"]))
        for o, d, expected in blk["cases"]:
            assert b.intersects(K.point(o), O.norm(K.vector(d))) == expected
    c = B["shape_in_parent_space"]
    b = O.Sphere(O.mat_mul(O.translation(1.0, -3.0, 5.0), O.scaling(0.5, 2.0, 4.0))).parent_space_bounding_box()
    K.assert_exact(b.min, K.point(c["min"]))
    K.assert_exact(b.max, K.point(c["max"]))
    c = B["cone_unbounded"]
    b = O.Cone().bounding_box()
    K.assert_exact(b.min, K.point(c["min"]))
    K.assert_exact(b.max, K.point(c["max"]))
    c = B["cone_bounded"]
    b = O.Cone(minimum_y=c["min_y"], maximum_y=c["max_y"]).bounding_box()
    K.assert_exact(b.min, K.point(c["min"]))
    K.assert_exact(b.max, K.point(c["max"]))


def test_group_transform_baking(kat):  # shape/group.rs:216-340
    G = kat["group"]
    g = O.GroupShape.with_children([O.Sphere(), O.Sphere(), O.Sphere()])
    g.set_material(O.Material(shininess=123.456))
    assert [c.shininess for c in g.get_children()] == [f32(123.456)] * 3
    assert O.GroupShape().local_intersect(O.point(0, 0, 0), O.vector(0, 0, 1)) == []
    c = G["nonempty_group"]
    s1, s2, s3 = O.Sphere(), O.Sphere(O.translation(0.0, 0.0, -3.0)), O.Sphere(O.translation(5.0, 0.0, 0.0))
    g = O.GroupShape()
    for s in (s1, s2, s3):
        g.add_child(s)
    xs = sorted(g.local_intersect(K.point(c["ray"][0]), K.vector(c["ray"][1])), key=lambda x: x[0])
    assert len(xs) == c["hits"]
    assert [x[1].node for x in xs] == [s2._node, s2._node, s1._node, s1._node]
    c = G["baked_child_transform"]
    expect = K.mat(c["expect_exact"])

    def check(g):
        K.assert_exact(g.get_children()[0].transformation(), expect)
        assert len(g.intersect(K.point(c["ray"][0]), K.vector(c["ray"][1]))) == c["hits"]
    g = O.GroupShape()
    g.set_transformation(O.scaling(2.0, 2.0, 2.0))
    g.add_child(O.Sphere(O.translation(5.0, 0.0, 0.0)))
    check(g)
    g = O.GroupShape()
    g.add_child(O.Sphere(O.translation(5.0, 0.0, 0.0)))
    g.set_transformation(O.scaling(2.0, 2.0, 2.0))
    check(g)
    g = O.GroupShape()
    g.set_transformation(O.scaling(3.0, 4.0, 8.0))
    g.add_child(O.Sphere(O.translation(5.0, 0.0, 0.0)))
    g.set_transformation(O.scaling(2.0, 2.0, 2.0))
    check(g)


def _nested(api):
    g1 = api.GroupShape()
    g1.set_transformation(api.rotation_y(K.CONSTS["PI"] / f32(2.0)))
    g2 = api.GroupShape()
    g2.set_transformation(api.scaling(1.0, 2.0, 3.0))
    g2.add_child(api.Sphere(api.translation(5.0, 0.0, 0.0)))
    g1.add_child(g2)
    return g1


def test_group_child_spaces_and_bounds(kat):  # shape/group.rs:342-455, shape/shape.rs:253-276
    G = kat["group"]
    s = _nested(O).get_children()[0].get_children()[0]
    K.assert_eps(s.world_to_object_point(K.point(G["world_to_object"]["point"])), G["world_to_object"]["expect_eps"])
    K.assert_eps(s.normal_at(K.point(G["normal_on_child"]["point"])), G["normal_on_child"]["expect_eps"])
    c = G["bounding_box_contains_children"]
    g = O.GroupShape()
    g.add_child(O.Sphere(O.mat_mul(O.translation(2.0, 5.0, -3.0), O.scaling(2.0, 2.0, 2.0))))
    g.add_child(O.Cylinder(O.mat_mul(O.translation(-4.0, -1.0, 4.0), O.scaling(0.5, 1.0, 0.5)), minimum_y=-2.0, maximum_y=2.0))
    b = g.bounding_box()
    K.assert_exact(b.min, K.point(c["min"]))
    K.assert_exact(b.max, K.point(c["max"]))
    g = O.GroupShape()
    g.add_child(O.Sphere(O.scaling(2.0, 2.0, 2.0)))
    g.add_child(O.Cylinder(O.scaling(2.0, 2.0, 2.0), minimum_y=-1.0, maximum_y=1.0))
    g.set_transformation(O.scaling(0.5, 0.5, 0.5))
    b1, b2 = g.bounding_box(), g.parent_space_bounding_box()
    assert np.array_equal(b1.min, b2.min) and np.array_equal(b1.max, b2.max)
    # the box gates the children (:432-455): a test shape's unit box, missed and hit
    g = O.GroupShape()
    g.add_child(O.Sphere())
    assert g.intersect(O.point(0, 0, -5), O.vector(0, 1, 0)) == []
    assert len(g.intersect(O.point(0, 0, -5), O.vector(0, 0, 1))) == 2


def test_group_divide(kat):  # shape/group.rs:458-639
    G = kat["group"]
    s1, s2, s3 = O.Sphere(O.translation(-2.0, -2.0, 0.0)), O.Sphere(O.translation(-2.0, 2.0, 0.0)), O.Sphere(O.scaling(4.0, 4.0, 4.0))
    g = O.GroupShape()
    for s in (s1, s2, s3):
        g.add_child(s)
    g.divide(1)
    ch = g.get_children()
    assert ch[0].node == s3._node and ch[1].is_group
    assert [c.node for c in ch[1].get_children()] == [s1._node, s2._node]
    # partitioning_children (:458-489) seen through divide(3): left / right singletons are pushed back as themselves
    s1, s2, s3 = O.Sphere(O.translation(-2.0, 0.0, 0.0)), O.Sphere(O.translation(2.0, 0.0, 0.0)), O.Sphere()
    g = O.GroupShape()
    for s in (s1, s2, s3):
        g.add_child(s)
    g.divide(3)
    assert [c.node for c in g.get_children()] == [s3._node, s1._node, s2._node]
    # subdividing_group_with_too_few_children (:551-604)
    s1, s2, s3, s4 = (O.Sphere(O.translation(-2.0, 0.0, 0.0)), O.Sphere(O.translation(2.0, 1.0, 0.0)),
                      O.Sphere(O.translation(2.0, -1.0, 0.0)), O.Sphere())
    sub = O.GroupShape()
    for s in (s1, s2, s3):
        sub.add_child(s)
    g = O.GroupShape()
    g.add_child(sub)
    g.add_child(s4)
    g.divide(3)
    ch = g.get_children()
    assert ch[0].node == sub.node and ch[1].node == s4._node
    sc = ch[0].get_children()
    assert sc[0].node == s1._node and [c.node for c in sc[1].get_children()] == [s2._node, s3._node]
    # divide_preserves_pushed_down_transformation (:607-639)
    c = G["divide_preserves_transformation"]
    group = O.GroupShape()
    group.set_transformation(O.translation(1.0, 1.0, 0.0))
    for t in ((-2.0, 0.0, 0.0), (2.0, -1.0, 0.0), (2.0, 1.0, 0.0)):
        group.add_child(O.Sphere(O.translation(*t)))
    group.divide(2)
    ch = group.get_children()
    K.assert_exact(ch[0].transformation(), O.translation(*c["s1"]))
    sub = ch[1].get_children()
    K.assert_exact(sub[0].transformation(), O.translation(*c["s2"]))
    K.assert_exact(sub[1].transformation(), O.translation(*c["s3"]))


def _default_triangle(kat, api=O):
    c = kat["triangle"]["default"]
    return api.Triangle(K.point(c["p1"]), K.point(c["p2"]), K.point(c["p3"]))


def test_triangle(kat):  # shape/triangle.rs:101-176
    T = kat["triangle"]
    t = _default_triangle(kat)
    e1, e2, normal = t.triangle_fields()
    K.assert_exact(e1, K.vector(T["construction"]["e1"]))
    K.assert_exact(e2, K.vector(T["construction"]["e2"]))
    K.assert_exact(normal, K.vector(T["construction"]["normal"]))
    for p in T["normal_points"]["points"]:
        K.assert_exact(t.local_norm_at(K.point(p)), normal)
    for o, d in T["misses"]["rays"]:
        assert t.local_intersect(K.point(o), K.vector(d)) == []
    ts = t.local_intersect(K.point(T["strikes"]["ray"][0]), K.vector(T["strikes"]["ray"][1]))
    assert ts == [f32(T["strikes"]["distance_exact"])]
    c = T["bounding_box"]
    b = O.Triangle(*[K.point(p) for p in c["points"]]).bounding_box()
    K.assert_exact(b.min, K.point(c["min"]))
    K.assert_exact(b.max, K.point(c["max"]))


def test_smooth_triangle(kat):  # shape/smooth_triangle.rs:71-107
    T = kat["triangle"]
    c, d = T["smooth"], T["default"]
    t = O.SmoothTriangle(K.point(d["p1"]), K.point(d["p2"]), K.point(d["p3"]), K.vector(c["n1"]), K.vector(c["n2"]), K.vector(c["n3"]))
    (hit,) = t.local_intersect_uv(K.point(c["uv_ray"][0]), K.vector(c["uv_ray"][1]))
    assert hit[1] == f32(c["u_exact"]) and hit[2] == f32(c["v_exact"])
    K.assert_eps(t.normal_at_uv(O.point(0, 0, 0), c["u_exact"], c["v_exact"]), c["interpolated_normal_eps"])
    # what a render sees: the hit object is the inner flat Triangle (smooth_triangle.rs:37-39)
    K.assert_exact(t.normal_at(O.point(0, 0.5, 0)), K.vector(T["construction"]["normal"]))


# ---------------------------------------------------------------- pattern/uv.rs + canvas_from_ppm
def _ppm_text(lines):
    return "\n".join("        " + ln if ln else ln for ln in lines)


def _cube_map(kat, api):
    c = kat["uv"]["cube_map"]
    faces = {f: api.AlignCheck(*[c["names"][n] for n in names]) for f, names in c["faces"].items()}
    return api.CubicMap(faces["front"], faces["back"], faces["left"], faces["right"], faces["up"], faces["down"])


def test_uv_patterns_and_mappings(kat):  # pattern/uv.rs:387-640
    U = kat["uv"]
    c = U["checkers"]
    p = O.UVCheckers(c["width"], c["height"], c["a"], c["b"])
    for u, v, expect in c["cases_exact"]:
        K.assert_exact(p.color_at(u, v), expect)
    for pt, eu, ev in U["spherical"]["cases_eps"]:
        u, v = O.point_to_uv(O.SphericalMap(), K.point(pt))
        K.assert_eps([u, v], [eu, ev])
    tm = O.TextureMap(O.UVCheckers(16.0, 8.0, (0, 0, 0), (1, 1, 1)), O.SphericalMap())
    for pt, expect in U["texture_map_spherical"]["cases_exact"]:
        K.assert_exact(tm.color_at_world(K.point(pt)), expect)
    for pt, eu, ev in U["planar"]["cases_exact"]:
        u, v = O.point_to_uv(O.PlanarMap(), K.point(pt))
        assert (u, v) == (f32(eu), f32(ev)), pt
    for pt, eu, ev in U["cylindrical"]["cases_eps"]:
        u, v = O.point_to_uv(O.CylindricalMap(), K.point(pt))
        K.assert_eps([u, v], [eu, ev])
    c = U["align_check"]
    ac = O.AlignCheck(*c["colors"])
    for u, v, idx in c["cases_exact"]:
        K.assert_exact(ac.color_at(u, v), c["colors"][idx])
    for pt, face in U["faces"]["cases"]:
        assert O.face_from_point(K.point(pt)) == face
    for face, pt, eu, ev in U["cube_uv"]["cases_exact"]:
        assert O.cube_uv(face, K.point(pt)) == (f32(eu), f32(ev)), (face, pt)
    cm = _cube_map(kat, O)
    for pt, name in U["cube_map"]["cases_exact"]:
        K.assert_exact(cm.color_at_world(K.point(pt)), U["cube_map"]["names"][name])


def test_ppm_reader_and_uv_image(kat):  # canvas.rs:281-398, pattern/uv.rs:642-669
    R = kat["ppm_reader"]
    with pytest.raises(O.PpmParseError) as e:
        O.canvas_from_ppm(_ppm_text(R["wrong_magic"]["lines"]))
    assert e.value.kind == R["wrong_magic"]["error"]
    img = O.canvas_from_ppm(_ppm_text(R["size"]["lines"]))
    assert img.shape == (R["size"]["height"], R["size"]["width"], 3)
    img = O.canvas_from_ppm(_ppm_text(R["pixels"]["lines"]))
    for x, y, expect in R["pixels"]["cases_eps"]:
        K.assert_eps(img[y, x], expect)
    for key in ("comments", "spanning", "empty_lines", "scale"):
        img = O.canvas_from_ppm(_ppm_text(R[key]["lines"]))
        for x, y, expect in R[key]["cases_exact"]:
            K.assert_exact(img[y, x], expect)
    c = kat["uv"]["image"]
    pattern = O.UVImage(O.canvas_from_ppm(_ppm_text(c["ppm_lines"])))
    for u, v, expect in c["cases_exact"]:
        K.assert_exact(pattern.color_at(u, v), expect)
