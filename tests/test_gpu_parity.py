"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
the CPU oracle on the same inputs.  Bar: every f32 channel value equal
(== semantics) and identical ray counts; quantised bytes / PPM text identical.

Sizes are chosen so the single-threaded oracle finishes in seconds; the
BASELINE.json full-size configurations are covered by sampled rows and by
size-independent properties in test_gpu_fullsize.py.
"""
import ctypes as C

import numpy as np
import pytest

import ray_tracer_challenge_amd as P
from oracle import oracle as O
from ray_tracer_challenge_amd import _lib as L
from ray_tracer_challenge_amd import scenes
from tests import helpers as H
from tests import kat as K

pytestmark = pytest.mark.gpu
f32 = np.float32
S2 = K.CONSTS["FRAC_1_SQRT_2"]


def _render_both(world, camera, depth, threads=8):
    canvas = camera.render(world, depth)
    img, rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=threads)
    return canvas, camera.last_stats, img, rays


# ----------------------------------------------------------------------- powf
def test_device_powf_equals_libm_powf():
    libm = C.CDLL("libm.so.6")
    libm.powf.restype = C.c_float
    libm.powf.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.random(60_000, dtype=f32), f32(1.0) - rng.random(20_000, dtype=f32) * f32(1e-3),
                         np.exp(rng.uniform(-60, 60, 20_000)).astype(f32),
                         np.array([1e-40, 1e-45, 1.0, 2.0, 0.5, np.inf], dtype=f32)])
    for y in (10.0, 50.0, 200.0, 300.0, 2.5, 0.0, 1.0, -3.0):
        ys = np.full(xs.shape, y, dtype=f32)
        got = P.powf(xs, ys)
        exp = np.array([libm.powf(float(x), float(y)) for x in xs], dtype=f32)
        same = (got == exp) | (np.isnan(got) & np.isnan(exp))
        assert same.all(), (y, xs[~same][:4], got[~same][:4], exp[~same][:4])
        assert np.array_equal(got, P.powf_host(xs, ys), equal_nan=True)


def test_device_cosf_equals_libm_cosf():
    """f32::cos (pattern/sine_2d.rs:40) on the device == glibc cosf == the host compile of the same code."""
    libm = C.CDLL("libm.so.6")
    libm.cosf.restype = C.c_float
    libm.cosf.argtypes = [C.c_float]
    rng = np.random.default_rng(5)
    xs = np.concatenate([
        rng.uniform(-1, 1, 40_000), rng.uniform(-130, 130, 80_000), rng.uniform(-1e6, 1e6, 40_000),
        rng.standard_normal(40_000) * 10.0 ** rng.uniform(-20, 38, 40_000),
        [0.0, -0.0, np.pi, np.pi / 2, np.pi / 4, 120.0, 119.99999, 1e-4, 2.0 ** -12, 3.4e38, -3.4e38, 1e-40, np.inf, np.nan],
    ]).astype(f32)
    xs = np.concatenate([xs, rng.integers(0, 2 ** 32, 100_000, dtype=np.uint64).astype(np.uint32).view(f32)])
    got = P.cosf(xs)
    exp = np.array([libm.cosf(float(v)) for v in xs], dtype=f32)
    same = (got.view(np.uint32) == exp.view(np.uint32)) | (np.isnan(got) & np.isnan(exp))
    assert same.all(), (xs[~same][:5], got[~same][:5], exp[~same][:5])
    assert np.array_equal(got, P.cosf_host(xs), equal_nan=True)


def test_exact_sqrt_and_divide_cores_equal_ieee_forms():
    """normalize_exact's cheap branch (sqrt_core / RcpCore) must return the same bits as sqrtf and '/'
    everywhere inside its validity range: 12 M random vectors across the whole exponent range it admits,
    plus vectors with zero / tiny / dominant components and exact squares."""
    rng = np.random.default_rng(2024)
    n = 4_000_000
    blocks = []
    for lo, hi in ((-38, 18), (-3, 3), (-12, -8)):
        mant = rng.uniform(1.0, 2.0, (n, 3))
        expo = rng.integers(lo, hi, (n, 3))
        sign = rng.choice([-1.0, 1.0], (n, 3))
        blocks.append((sign * mant * np.exp2(expo)).astype(f32))
    edge = rng.uniform(-4, 4, (200_000, 3)).astype(f32)
    edge[::4, 0] = 0.0
    edge[1::4, 1] = 0.0
    edge[2::4, 2] = f32(1e-25)
    edge[3::4] = np.round(edge[3::4])  # small integers: exact squares, exact quotients
    edge[5::8, 1:] = 0.0
    blocks.append(edge)
    total_in, total_bad = 0, 0
    for v in blocks:
        v = np.ascontiguousarray(v)
        counts = (C.c_uint32 * 2)()
        L.check(P.lib().rtc_selftest_fastmath(v.ctypes.data_as(L.FP), v.shape[0], 0, counts))
        total_in += counts[0]
        total_bad += counts[1]
    assert total_bad == 0, (total_bad, total_in)
    assert total_in > 11_000_000, total_in


# ------------------------------------------- the reference's world.rs tests, on device
def _both_color_at(pw, o, d, depth):
    o, d = np.asarray(o, dtype=f32), np.asarray(d, dtype=f32)
    got = pw.color_at(o.reshape(1, 4), d.reshape(1, 4), depth)[0]
    exp = H.oracle_world(pw).color_at(o, d, depth)
    K.assert_exact(got, exp)
    return got


def test_color_at_known_answers(kat):  # world.rs:563-590
    W = kat["world"]
    w = P.default_world()
    c = W["color_when_ray_hits"]
    got = _both_color_at(w, K.point(c["ray"][0]), K.vector(c["ray"][1]), c["depth"])
    K.assert_eps(got, c["expect_eps"])
    c = W["color_when_ray_misses"]
    K.assert_exact(_both_color_at(w, K.point(c["ray"][0]), K.vector(c["ray"][1]), c["depth"]), c["expect_exact"])
    c = W["color_when_intersection_behind_ray"]
    w.objects[0].material = P.Material(ambient=1.0)
    w.objects[1].material = P.Material(ambient=1.0)
    K.assert_exact(_both_color_at(w, K.point(c["ray"][0]), K.vector(c["ray"][1]), c["depth"]), c["expect_exact"])


def test_shading_scenarios_match_oracle(kat):  # world.rs:483-537, 646-658, 747-844 (as color_at rays)
    o, d = P.point(0, 0, -3), np.array([0, -S2, S2, 0], dtype=f32)
    w = P.default_world()
    w.objects.append(P.Plane(P.translation(0.0, -1.0, 0.0), P.Material(reflective=0.5)))
    got = _both_color_at(w, o, d, 1)
    # the reference feeds shade_hit the literal t = SQRT_2; color_at finds t itself, 1 ulp away at most
    K.assert_eps(got, kat["world"]["shade_hit_reflective"]["expect_eps"], eps=f32(4e-6))
    _both_color_at(w, o, d, 0)
    for floor in (P.Material(transparency=0.5, refractive_index=1.5),
                  P.Material(reflective=0.5, transparency=0.5, refractive_index=1.5)):
        w = P.default_world()
        w.objects.append(P.Plane(P.translation(0.0, -1.0, 0.0), floor))
        w.objects.append(P.Sphere(P.translation(0.0, -3.5, -0.5), P.Material(color=(1, 0, 0), ambient=0.5)))
        for depth in (5, 1, 0):
            _both_color_at(w, o, d, depth)
    # shade_hit_for_intersection_in_shadow
    w = P.World([P.Sphere(), P.Sphere(P.translation(0.0, 0.0, 10.0))], P.PointLight(P.point(0, 0, -10), P.color(1, 1, 1)))
    K.assert_exact(_both_color_at(w, P.point(0, 0, 5), P.vector(0, 0, 1), 1), K.vec([0.1, 0.1, 0.1]))
    # shade_hit_with_mutually_reflective_surfaces terminates (world.rs:511-523)
    m = P.Material(reflective=1.0)
    w = P.World([P.Plane(P.translation(0.0, -1.0, 0.0), m), P.Plane(P.translation(0.0, 1.0, 0.0), m)],
                P.PointLight(P.point(0, 0, 0), P.color(0, 0, 0)))
    _both_color_at(w, P.point(0, 0, 0), P.vector(0, 1, 0), 1)
    _both_color_at(w, P.point(0, 0, 0), P.vector(0, 1, 0), L.RTC_STACK_DEPTH_BASE)


def test_nested_glass_refraction_indices_match_oracle(kat):  # world.rs:396-451 geometry, traced end to end
    def glass(t, ri):
        return P.Sphere(t, P.Material(transparency=1.0, refractive_index=ri, diffuse=0.3, reflective=0.2))
    w = P.World([glass(P.scaling(2.0, 2.0, 2.0), 1.5), glass(P.translation(0.0, 0.0, -0.25), 2.0),
                 glass(P.translation(0.0, 0.0, 0.25), 2.5),
                 P.Plane(P.translation(0.0, -3.0, 0.0), P.Material(color=(1, 0.5, 0.2)))],
                P.PointLight(P.point(-10, 10, -10), P.color(1, 1, 1)))
    rng = np.random.default_rng(3)
    n = 256
    o = np.zeros((n, 4), dtype=f32)
    o[:, :3] = rng.uniform(-0.3, 0.3, (n, 3)) + np.array([0, 0, -4])
    o[:, 3] = 1
    d = np.zeros((n, 4), dtype=f32)
    d[:, :3] = rng.uniform(-0.25, 0.25, (n, 3)) + np.array([0, 0, 1])
    d = np.stack([O.norm(v) for v in d]).astype(f32)
    got = w.color_at(o, d, 5)
    ow = H.oracle_world(w)
    exp = np.stack([ow.color_at(o[i], d[i], 5) for i in range(n)])
    H.assert_images_equal(got.reshape(n, 1, 3), exp.reshape(n, 1, 3), "nested glass")
    # rays starting inside the spheres exercise the negative-t container walk
    o[:, :3] = rng.uniform(-0.6, 0.6, (n, 3))
    got = w.color_at(o, d, 5)
    exp = np.stack([ow.color_at(o[i], d[i], 5) for i in range(n)])
    H.assert_images_equal(got.reshape(n, 1, 3), exp.reshape(n, 1, 3), "nested glass, inside")


def test_is_shadowed_and_point_light_intensity(kat):  # world.rs:593-630
    w = P.default_world()
    c = kat["world"]["is_shadowed"]
    pts = np.stack([K.point(p) for p, _ in c["cases"]])
    lights = np.tile(K.point(c["light_position"]), (len(pts), 1))
    assert list(w.is_shadowed(lights, pts)) == [e for _, e in c["cases"]]
    c = kat["world"]["point_light_intensity_at"]
    pts = np.stack([K.point(p) for p, _ in c["cases"]])
    K.assert_exact(w.intensity_at(pts), [e for _, e in c["cases"]])


def test_rectangle_light_intensity_matches_oracle():  # rectangle_light.rs:142-166 geometry
    rng = np.random.default_rng(11)
    pts = np.ones((300, 4), dtype=f32)
    pts[:, :3] = rng.uniform(-2.5, 2.5, (300, 3))
    for jitter in (("constant", 0.5), ("constant", 0.25), ("hashed", 99)):
        w = P.default_world()
        w.light = P.RectangleLight(P.color(1, 1, 1), P.point(-0.5, -0.5, -5), P.vector(1, 0, 0), 2, P.vector(0, 1, 0), 2,
                                   jitter)
        got = w.intensity_at(pts)
        ow = H.oracle_world(w)
        exp = []
        for i, p in enumerate(pts):
            ow.set_pixel(i)
            exp.append(ow.intensity_at(p))
        assert np.array_equal(got, np.array(exp, dtype=f32)), jitter
    assert set(np.unique(got)).issubset({0.0, 0.25, 0.5, 0.75, 1.0})


# ------------------------------------------------------------- whole renders
def test_render_world_known_answer(kat):  # camera.rs:156-167
    c = kat["camera"]["render_world"]
    cam = P.Camera(*c["size"], K.CONSTS["PI"] / f32(2.0),
                   P.view_transform(K.point(c["from"]), K.point(c["to"]), K.vector(c["up"])))
    canvas, stats, img, rays = _render_both(P.default_world(), cam, c["depth"], threads=1)
    K.assert_eps(canvas.pixel_at(*c["pixel"]), c["expect_eps"])
    H.assert_images_equal(canvas.data, img, "render_world")
    assert stats["rays"] == rays and stats["pixels"] == 100


@pytest.mark.parametrize("name,size,kw", [
    ("single_sphere", (96, 96), {}),
    ("soft_shadows", (100, 40), {"jitter": ("constant", 0.5)}),
    ("soft_shadows", (100, 40), {"jitter": ("hashed", scenes.DEFAULT_SEED)}),
    ("soft_shadows", (61, 67), {"jitter": ("hashed", 12345)}),
    ("first_scene", (100, 50), {}),
    ("first_plane", (100, 50), {}),
    ("glass_and_mirror", (96, 96), {}),
    ("sphere_grid", (128, 128), {}),
    ("shapes_medley", (128, 96), {}),
    ("shapes_medley", (64, 48), {"jitter": ("constant", 0.5)}),
    ("first_patterns", (100, 50), {}),
    ("first_patterns", (320, 160), {}),
    ("reflect_refract", (200, 100), {}),
    ("patterns_medley", (128, 96), {}),
    ("patterns_medley", (64, 48), {"jitter": ("constant", 0.5)}),
])
def test_render_matches_oracle_bitwise(name, size, kw):
    world, camera, depth = getattr(scenes, name)(*size, **kw)
    canvas, stats, img, rays = _render_both(world, camera, depth)
    H.assert_images_equal(canvas.data, img, name)
    assert stats["rays"] == rays, (stats["rays"], rays)
    assert stats["pixels"] == (size[0] - 1) * (size[1] - 1)
    # camera.rs:80-81: last row / column never traced
    assert not canvas.data[-1].any() and not canvas.data[:, -1].any()
    # canvas.rs:39-96: the wire format
    assert np.array_equal(O.quantize(canvas.data), O.quantize(img))
    assert canvas.to_ppm() == O.to_ppm(img)


@pytest.mark.parametrize("depth", [0, 1, 2, 3, 8])
def test_recursion_depths(depth):
    world, camera, _ = scenes.glass_and_mirror(64, 64)
    canvas, stats, img, rays = _render_both(world, camera, depth)
    H.assert_images_equal(canvas.data, img, "depth %d" % depth)
    assert stats["rays"] == rays


def test_degenerate_sizes_and_empty_world():
    world, _, depth = scenes.single_sphere(8, 8)
    for (w, h) in [(1, 1), (1, 7), (9, 1), (2, 2), (17, 3)]:
        cam = P.Camera(w, h, scenes.PI / f32(3.0), P.view_transform(P.point(0, 0, -5), P.point(0, 0, 0), P.vector(0, 1, 0)))
        canvas, stats, img, rays = _render_both(world, cam, depth, threads=1)
        H.assert_images_equal(canvas.data, img, "%dx%d" % (w, h))
        assert stats["rays"] == rays and stats["pixels"] == (w - 1) * (h - 1)
    empty = P.World([], P.PointLight(P.point(0, 0, 0), P.color(1, 1, 1)))
    cam = P.Camera(16, 16, scenes.PI / f32(3.0), P.identity_4x4())
    canvas = cam.render(empty, 5)
    assert not canvas.data.any() and cam.last_stats["rays"] == 15 * 15


def test_boundary_errors_on_device():
    w = P.default_world()
    cam = P.Camera(8, 8, 1.0, P.identity_4x4())
    with pytest.raises(P.RtcError) as e:
        cam.render(w, L.RTC_MAX_DEPTH + 1)
    assert e.value.status == L.RTC_ERR_INVALID_ARG
    with pytest.raises(P.RtcError) as e:
        cam.render(w, -1)
    assert e.value.status == L.RTC_ERR_INVALID_ARG
    w.light = None
    with pytest.raises(P.RtcError) as e:
        cam.render(w, 5)
    assert e.value.status == L.RTC_ERR_NO_LIGHT
    proj = np.eye(4, dtype=f32)
    proj[3, 2] = 0.5  # projective transform: inverse's last row is not [0,0,0,1]
    w = P.World([P.Sphere(proj)], P.PointLight(P.point(0, 0, 0), P.color(1, 1, 1)))
    with pytest.raises(P.RtcError) as e:
        cam.render(w, 5)
    assert e.value.status == L.RTC_ERR_UNSUPPORTED


# ------------------------------------------------- scene-specialised (hiprtc) kernels
@pytest.mark.parametrize("name,size,kw", [
    ("soft_shadows", (100, 40), {"jitter": ("hashed", scenes.DEFAULT_SEED)}),
    ("soft_shadows", (64, 64), {"jitter": ("constant", 0.5)}),
    ("single_sphere", (64, 64), {}),
    ("first_scene", (100, 50), {}),
    ("first_plane", (100, 50), {}),
    ("glass_and_mirror", (96, 96), {}),
    ("shapes_medley", (128, 96), {}),
    ("first_patterns", (100, 50), {}),
    ("reflect_refract", (160, 80), {}),
    ("patterns_medley", (128, 96), {}),
    ("sphere_grid", (128, 128), {}),          # 64 like objects: the any-count loop specialised on their shared flags word
    ("sphere_grid", (96, 64), {"n": 5}),      # 25 objects: the remainder loop
    ("hexagons", (128, 64), {}),              # GroupShape trees: the traversal kernel compiled for the scene
    ("groups_medley", (96, 64), {}),
    ("grouped_grid", (96, 96), {}),
    ("mesh", (64, 64), {}),                   # triangles only: their flags word is a compile-time constant
    ("first_textures", (96, 48), {}),
    ("skybox", (64, 64), {}),
])
def test_specialised_kernel_matches_generic_and_oracle(name, size, kw, monkeypatch):
    """RTC_AMD_SPECIALIZE=1 compiles the kernel for this scene's shape with hiprtc; the image, the ray
    count and the shaded-hit count must equal the ahead-of-time kernel's and the oracle's."""
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = getattr(scenes, name)(*size, **kw)
    monkeypatch.setenv("RTC_AMD_BVH", "0")  # sphere_grid would otherwise take the tree kernel (test_flat_bvh.py)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RTC_AMD_SPECIALIZE", mode)
        r = Renderer(world, camera, device=0)
        assert r.kernel_name.startswith("render_kernel_spec[" if mode == "1" else "render_kernel<"), r.kernel_name
        img = r.render(depth).cpu().numpy()
        out[mode] = (img, r.stats())
        r.close()
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=8)
    for mode in ("0", "1"):
        H.assert_images_equal(out[mode][0], exp, "%s specialise=%s" % (name, mode))
        assert out[mode][1]["rays"] == rays
    assert out["0"][1]["shaded_hits"] == out["1"][1]["shaded_hits"]


def test_specialisation_policy_defaults(monkeypatch):
    from ray_tracer_challenge_amd.renderer import Renderer
    monkeypatch.delenv("RTC_AMD_SPECIALIZE", raising=False)
    world, camera, _ = scenes.soft_shadows(64, 64)
    assert Renderer(world, camera, device=0).kernel_name == "render_kernel<4,simple>"      # thumbnail: AOT
    world, camera, _ = scenes.soft_shadows(1024, 512)
    assert Renderer(world, camera, device=0).kernel_name.startswith("render_kernel_spec[")  # >= 2^18 pixels
    world, camera, _ = scenes.sphere_grid(1024, 512)
    assert Renderer(world, camera, device=0).kernel_name == "render_kernel_spec[tree,bvh;all 0x500]"  # 64 bounded objects
    monkeypatch.setenv("RTC_AMD_BVH", "0")
    assert Renderer(world, camera, device=0).kernel_name == "render_kernel_spec[all 0x500]"  # 64 like objects
    world.objects[3].casts_shadow = False                                                  # ... no longer alike
    assert Renderer(world, camera, device=0).kernel_name == "render_kernel<0,general>"


# ------------------------------------------------- shapes and patterns on the device
def _kinds():
    return {"sphere": (P.Sphere, O.Sphere), "plane": (P.Plane, O.Plane), "cube": (P.Cube, O.Cube),
            "cylinder": (P.Cylinder, O.Cylinder), "cone": (P.Cone, O.Cone)}


def test_cone_known_answers_on_device(kat):  # shape/cone.rs:190-278 through rtc_local_intersect / rtc_normal_at
    co = kat["cone"]
    c = P.Cone()
    rays = [(K.point(o), O.norm(K.vector(d))) for o, d, _, _ in co["sides"]["cases"]]
    got = c.local_intersect([r[0] for r in rays], [r[1] for r in rays])
    for ts, (_, _, t0, t1) in zip(got, co["sides"]["cases"]):
        assert len(ts) == 2
        K.assert_eps(ts, [t0, t1])
    blk = co["parallel_to_one_half"]
    (ts,) = c.local_intersect([K.point(blk["ray"][0])], [O.norm(K.vector(blk["ray"][1]))])
    assert len(ts) == 1
    K.assert_eps(ts, blk["expect_eps"])
    blk = co["caps"]
    cc = P.Cone(minimum_y=blk["min"], maximum_y=blk["max"], closed=blk["closed"])
    got = cc.local_intersect([K.point(o) for o, _, _ in blk["cases"]], [O.norm(K.vector(d)) for _, d, _ in blk["cases"]])
    assert [len(ts) for ts in got] == [n for _, _, n in blk["cases"]]
    # local_norm_at through normal_at on an untransformed cone: the expected vectors, normalised (shape.rs:72-154)
    pts = [K.point(p) for p, _ in co["normals"]["cases"]]
    got = c.normal_at(pts)
    exp = O.Cone().normal_at
    for g, p in zip(got, pts):
        e = exp(p)
        assert np.array_equal(g, e, equal_nan=True), (p, g, e)


@pytest.mark.parametrize("kind", ["sphere", "plane", "cube", "cylinder", "cone"])
def test_local_intersect_and_normal_at_match_oracle_bitwise(kind):
    """Shape::local_intersect / normal_at for random rays and points, bounded / closed variants included:
    the same distances in the same push order, bit for bit."""
    rng = np.random.default_rng(sorted(_kinds()).index(kind) + 3)
    pk, ok = _kinds()[kind]
    variants = [dict()]
    if kind in ("cylinder", "cone"):
        variants += [dict(minimum_y=-0.5, maximum_y=0.75, closed=True), dict(minimum_y=0.0, maximum_y=1.5, closed=False),
                     dict(minimum_y=-2.0, maximum_y=-0.25, closed=True)]
    t = P.chain(P.translation(0.3, -0.2, 0.1), P.rotation_z(f32(0.4)), P.scaling(0.8, 1.3, 0.6))
    n = 4000
    o = np.concatenate([rng.uniform(-3, 3, (n, 3)), np.ones((n, 1))], axis=1).astype(f32)
    d = np.concatenate([rng.standard_normal((n, 3)), np.zeros((n, 1))], axis=1).astype(f32)
    d[: n // 8, 1] = 0.0                      # rays parallel to the caps / the plane
    d[n // 8: n // 4, 0] = d[n // 8: n // 4, 2] = 0.0    # rays along the axis
    d[n // 4: n // 4 + 200, 1] = np.hypot(d[n // 4: n // 4 + 200, 0], d[n // 4: n // 4 + 200, 2])  # cone: parallel to one half
    o[n // 2: n // 2 + 300] = [0.0, 0.0, 0.0, 1.0]       # through the apex
    d = np.array([O.norm(v) for v in d], dtype=f32)
    d[~np.isfinite(d).all(axis=1)] = [0.0, 1.0, 0.0, 0.0]
    for kw in variants:
        ps, os_ = pk(t, None, **kw), ok(t, None, **kw)
        got = ps.local_intersect(o, d)
        hits = 0
        for i in range(n):
            exp = os_.local_intersect(o[i], d[i])
            assert len(got[i]) == len(exp), (kind, kw, i, o[i], d[i], got[i], exp)
            assert np.array_equal(np.array(got[i], dtype=f32), np.array(exp, dtype=f32), equal_nan=True), (kind, kw, i, got[i], exp)
            hits += len(exp)
        assert kind == "plane" or hits > n // 8
        pts = np.concatenate([rng.uniform(-2, 2, (600, 3)), np.ones((600, 1))], axis=1).astype(f32)
        gn = ps.normal_at(pts)
        for i in range(600):
            en = os_.normal_at(pts[i])
            assert np.array_equal(gn[i], en, equal_nan=True), (kind, kw, pts[i], gn[i], en)


PATTERNS = {"stripes": (P.Stripes, O.Stripes), "gradient": (P.Gradient, O.Gradient), "rings": (P.Rings, O.Rings),
            "checkers": (P.Checkers, O.Checkers), "sine_2d": (P.Sine2D, O.Sine2D)}


def test_pattern_known_answers_on_device(kat):  # pattern/*.rs tests through rtc_pattern_color_at
    for name, (pp, _) in PATTERNS.items():
        blk = kat["pattern"][name]
        pat = pp(blk["a"], blk["b"])
        pts = [K.point(p) for p, _ in blk["cases_exact"]]
        got = pat.color_at_world(pts)
        for g, (_, expect) in zip(got, blk["cases_exact"]):
            K.assert_exact(g, K.vec(expect))
        for p, expect in blk.get("cases_eps", []):
            K.assert_eps(pat.color_at_world([K.point(p)])[0], K.vec(expect))
    # phong_lighting.rs:197-235: ambient-only material, so the lit colour is the pattern colour
    c = kat["pattern"]["phong_with_pattern"]
    got = P.Stripes().color_at_object([K.point(p) for p, _ in c["cases_exact"]], P.Sphere())
    for g, (_, expect) in zip(got, c["cases_exact"]):
        K.assert_exact(g, expect)


@pytest.mark.parametrize("name", sorted(PATTERNS))
def test_pattern_color_at_object_matches_oracle_bitwise(name):
    """Pattern::color_at_object (pattern.rs:15-19) with object and pattern transforms, for points near and far
    (large pattern-space coordinates take cosf's big-argument reduction and the saturating `as i32`)."""
    rng = np.random.default_rng(len(name))
    pp, op = PATTERNS[name]
    a, b = (0.1, 1.0, 0.5), (0.9, 0.2, 0.6)
    pt = P.chain(P.scaling(0.005, 1.0, 0.005), P.translation(-5.0, 1.0, 0.5), P.rotation_y(f32(0.3)))
    ot = P.chain(P.shearing(0.0, 1.0, 0.0, 0.0, 0.0, 1.0), P.translation(1.5, 0.5, -0.5), P.scaling(0.5, 0.5, 0.5))
    pts = np.concatenate([
        rng.uniform(-4, 4, (3000, 3)), rng.uniform(-300, 300, (2000, 3)), rng.standard_normal((1000, 3)) * 1e7,
        rng.standard_normal((200, 3)) * 1e15, np.zeros((1, 3))]).astype(f32)
    # (finite points only: the kernel skips products with a transform's structural zeros, which is exact for
    # finite operands -- see DESIGN.md "Arithmetic contract" -- but 0 * inf is NaN in the reference)
    pts = np.concatenate([pts, np.ones((pts.shape[0], 1), dtype=f32)], axis=1)
    for ptx, otx in ((None, None), (pt, None), (None, ot), (pt, ot)):
        got = pp(a, b, ptx).color_at_object(pts, P.Sphere(otx))
        opat, osh = op(a, b, ptx), O.Sphere(otx)
        for i in range(pts.shape[0]):
            exp = opat.color_at_object(pts[i], osh)
            assert np.array_equal(got[i], exp, equal_nan=True), (name, ptx is not None, otx is not None, pts[i], got[i], exp)


def test_pattern_boundary_errors_on_device():
    cam = P.Camera(8, 8, 1.0, P.identity_4x4())
    proj = np.eye(4, dtype=f32)
    proj[3, 0] = 0.25
    w = P.World([P.Sphere(None, P.Material(pattern=P.Stripes(transform=proj)))], P.PointLight(P.point(0, 0, -5), P.color(1, 1, 1)))
    with pytest.raises(P.RtcError) as e:
        cam.render(w, 5)
    assert e.value.status == L.RTC_ERR_UNSUPPORTED
    bad = P.Pattern(77)
    with pytest.raises(P.RtcError) as e:
        bad.color_at_world([P.point(0, 0, 0)])
    assert e.value.status == L.RTC_ERR_UNSUPPORTED


# ------------------------------------------------- sample-parallel rendering (2^s lanes per pixel share the light's cells)
@pytest.mark.parametrize("share", ["0", "1", "2", "3"])
@pytest.mark.parametrize("name,size,kw", [
    ("soft_shadows", (101, 37), {"jitter": ("hashed", scenes.DEFAULT_SEED)}),   # sizes that leave partial tiles
    ("soft_shadows", (45, 67), {"jitter": ("constant", 0.5)}),
    ("shapes_medley", (96, 72), {}),
    ("first_textures", (90, 50), {}),                                            # gates + patterns
    ("groups_medley", (80, 60), {}),                                             # traversal kernel
])
def test_lanes_sharing_a_pixel_change_nothing(name, size, kw, share, monkeypatch):
    """RTC_AMD_SHARE_LOG2=s: every pixel is traced by 2^s adjacent lanes that split each shade point's light cells
    between them (how small frames under an area light fill the chip).  Image, ray count and shaded-hit count must be
    those of one lane per pixel -- and of the oracle."""
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = getattr(scenes, name)(*size, **kw)
    monkeypatch.setenv("RTC_AMD_SPECIALIZE", "1")  # lane sharing is compiled into the per-scene kernels only
    monkeypatch.setenv("RTC_AMD_SHARE_LOG2", share)
    r = Renderer(world, camera, device=0)
    assert r.kernel_name.startswith("render_kernel_spec["), r.kernel_name
    img = r.render(depth).cpu().numpy()
    st = r.stats()
    r.close()
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=8)
    H.assert_images_equal(img, exp, "%s share=%s" % (name, share))
    assert st["rays"] == rays
    monkeypatch.setenv("RTC_AMD_SHARE_LOG2", "0")
    r0 = Renderer(world, camera, device=0)
    r0.render(depth)
    # (culled_shadow_rays may differ: the cull is decided per wave, and a wave now holds other pixels)
    assert r0.stats()["shaded_hits"] == st["shaded_hits"]
    r0.close()


# ------------------------------------------------- leaf-sharing tree walk + block lists (worlds with divided meshes)
@pytest.mark.parametrize("variant", ["default", "share0", "share1", "share2", "share3", "blocks_s0", "blocks_s1", "blocks_s3", "no_block_list",
                                     "image_order", "glass_s3", "nodes", "nodes_share0", "nodes_of_2_s1", "nodes_of_16_s3"])
@pytest.mark.parametrize("name,size", [("mesh", (230, 170)), ("here_be_dragons", (250, 100)), ("mesh", (64, 610))])
def test_lanes_splitting_leaf_runs_change_nothing(name, size, variant, monkeypatch, request):
    """Worlds whose GroupShapes hold long runs of leaves (the rings divide() leaves around a mesh): the 2^s lanes of a
    pixel split every run between them (for_each_leaf_shared: nearest hit, point-light shadow ray, the n1/n2 container
    walk), mesh runs take the ray into object space once, and the launch is a block list -- the tiles a mesh projects to
    first and with more lanes per pixel than the rest.  None of it may change a pixel, the ray count or the shaded hits."""
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = getattr(scenes, name)(*size)
    env = {"default": {}, "share0": {"RTC_AMD_SHARE_LOG2": "0"}, "share1": {"RTC_AMD_SHARE_LOG2": "1"}, "share2": {"RTC_AMD_SHARE_LOG2": "2"},
           "share3": {"RTC_AMD_SHARE_LOG2": "3"}, "blocks_s0": {"RTC_AMD_BLOCK_S": "0"}, "blocks_s1": {"RTC_AMD_BLOCK_S": "1"},
           "blocks_s3": {"RTC_AMD_BLOCK_S": "3"}, "no_block_list": {"RTC_AMD_BLOCK_LIST": "0"},
           "image_order": {"RTC_AMD_BLOCK_ORDER": "0"}, "glass_s3": {"RTC_AMD_BLOCK_S": "1", "RTC_AMD_BLOCK_S_TOP": "3"},
           # the library's own nodes over the long runs (cluster_leaf_runs; off by default in frames this small)
           "nodes": {"RTC_AMD_CLUSTERS": "1"}, "nodes_share0": {"RTC_AMD_CLUSTERS": "1", "RTC_AMD_SHARE_LOG2": "0"},
           "nodes_of_2_s1": {"RTC_AMD_CLUSTERS": "1", "RTC_AMD_CLUSTER_LEAF": "2", "RTC_AMD_CLUSTER_MIN_RUN": "4", "RTC_AMD_CLUSTER_GMAX": "0.99",
                             "RTC_AMD_BLOCK_S": "1"},
           "nodes_of_16_s3": {"RTC_AMD_CLUSTERS": "1", "RTC_AMD_CLUSTER_LEAF": "16", "RTC_AMD_BLOCK_S": "3"}}[variant]
    from tests.conftest import DEV_ONLY_SWITCHES
    if any(k in DEV_ONLY_SWITCHES for k in env):
        request.getfixturevalue("dev_lib")  # tuning constants are pinned through the development build only
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r = Renderer(world, camera, device=0)
    assert r.kernel_name.startswith("render_kernel_spec[tree"), r.kernel_name   # compiled for the scene whatever the size
    img = r.render(depth).cpu().numpy()
    st = r.stats()
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=8)
    H.assert_images_equal(img, exp, "%s %s" % (name, variant))
    assert st["rays"] == rays
    # a band partition (the multi-GPU split) of the same frame: block lists are built per partition
    parts = []
    for p in range(3):
        part = Renderer.partition(16, 3, p)
        parts.append(r.render(depth, part=part).cpu().numpy())
    got = np.zeros_like(img)
    cursor = [0, 0, 0]
    for b in range((camera.height + 15) // 16):
        p, y0, y1 = b % 3, b * 16, min((b + 1) * 16, camera.height)
        got[y0:y1] = parts[p][cursor[p]:cursor[p] + (y1 - y0)]
        cursor[p] += y1 - y0
    assert np.array_equal(got, img)
    r.close()


@pytest.mark.parametrize("name", ["mesh", "here_be_dragons"])
def test_nodes_at_a_frame_size_that_builds_them_by_default(name, monkeypatch):
    """Frames of more than 40 k waves are traced by two lanes per pixel in the mesh tiles and get the library's own nodes
    over their long triangle runs by default (rtc_device.hip: clusters_pay).  The small frames above force them on and
    compare with the oracle; here the default policy itself runs and is compared with the same frame without nodes."""
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = getattr(scenes, name)(2048, 1296)
    out = {}
    for mode in ("default", "0"):
        if mode == "default":
            monkeypatch.delenv("RTC_AMD_CLUSTERS", raising=False)
        else:
            monkeypatch.setenv("RTC_AMD_CLUSTERS", mode)
        r = Renderer(world, camera, device=0)
        img = r.render(depth).cpu().numpy()
        st = r.stats()
        out[mode] = (img, st["rays"], st["shaded_hits"])
        r.close()
    assert np.array_equal(out["default"][0].view(np.uint32), out["0"][0].view(np.uint32))
    assert out["default"][1:] == out["0"][1:]
    # row ranges against the oracle (through the middle of the meshes, and the frame's edges)
    oc, ow = H.oracle_camera(camera), H.oracle_world(world)
    for y0, y1 in ((0, 4), (560, 572), (700, 708), (1290, 1296)):
        exp, _ = oc.render(ow, depth, threads=8, rows=(y0, y1))
        H.assert_images_equal(out["default"][0][y0:y1], exp[y0:y1], "%s rows %d..%d" % (name, y0, y1))


def test_renderer_checks_the_buffers_it_hands_to_kernels():
    """Renderer passes raw device pointers on: a buffer of the wrong size, type or place is refused (ValueError, also under
    `python -O`), not written through."""
    import torch
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = scenes.single_sphere(64, 48)
    r = Renderer(world, camera, device=0)
    good = r.alloc()
    for bad in (torch.empty((47, 64, 3), dtype=torch.float32, device="cuda:0"), torch.empty((48, 64, 3), dtype=torch.float16, device="cuda:0"),
                torch.empty((48, 64, 3), dtype=torch.float32), good.permute(1, 0, 2)):
        with pytest.raises(ValueError):
            r.render(depth, out=bad)
    with pytest.raises(ValueError):
        r.quantize(good, out=torch.empty(good.numel() - 1, dtype=torch.uint8, device="cuda:0"))
    with pytest.raises(ValueError):
        r.to_ppm(good.cpu())
    img = r.render(depth, out=good).cpu().numpy()  # (and the good one still renders)
    exp, _ = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=4)
    H.assert_images_equal(img, exp, "single_sphere 64x48")
    r.close()
