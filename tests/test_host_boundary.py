"""CPU-only checks of the product's host side and its C-ABI boundary:
 * librtc_amd.so loads and exports every symbol include/rtc.h declares;
 * the flattened scene inputs (inverse transforms, camera, light cells) are
   bit-identical to the oracle's restatement of the reference's arithmetic;
 * the device powf restatement (host compile) equals the C library's powf --
   the routine f32::powf resolves to for a Linux build of the reference;
 * to_ppm is byte-identical to the oracle's Canvas::to_ppm;
 * error behaviour of the boundary (no light, bad kinds, closures, no GPU).
No compute call touches a GPU here.
"""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import ray_tracer_challenge_amd as P
from oracle import oracle as O
from ray_tracer_challenge_amd import _lib as L
from ray_tracer_challenge_amd import scenes
from tests import helpers as H

f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rtc.h")).read()
    body = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rtc_[a-z0-9_]+)\s*\(", body))
    assert len(declared) >= 40
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    lib = C.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert P.lib().rtc_abi_version() == 8


@pytest.mark.parametrize("name", ["soft_shadows", "first_scene", "first_plane", "glass_and_mirror", "shapes_medley",
                                  "single_sphere", "first_patterns", "reflect_refract", "patterns_medley"])
def test_flattened_scene_matches_oracle_bitwise(name):
    world, camera, _ = getattr(scenes, name)(64, 48)
    ow = H.oracle_world(world)
    for i, s in enumerate(world.objects):
        inv, _ = ow.shape_inverse(i)
        assert np.array_equal(s.transformation_inverse(), inv), (name, i)
        assert np.all(inv[3] == np.array([0, 0, 0, 1], dtype=f32))
        if s.material.pattern is not None:  # pattern.rs:52-54: the stored inverse, same cofactor arithmetic
            pinv = s.material.pattern.transformation_inverse()
            assert np.array_equal(pinv, O.inverse(s.material.pattern.transform)), (name, i)
            assert np.all(pinv[3] == np.array([0, 0, 0, 1], dtype=f32))
    oc = H.oracle_camera(camera)
    assert camera.pixel_size == oc.pixel_size
    assert camera.half_width == oc.half_width and camera.half_height == oc.half_height
    assert np.array_equal(camera.transform_inverse, oc.transform_inverse)
    for (x, y) in [(0, 0), (63, 47), (31, 7)]:
        po, pd = camera.ray_for_pixel(x, y)
        oo, od = oc.ray_for_pixel(x, y)
        assert np.array_equal(po, oo) and np.array_equal(pd, od)
    if hasattr(world.light, "corner"):
        pos, u, v, cells = ow.light_info()
        assert np.array_equal(world.light.position, pos)
        assert np.array_equal(world.light.cell_u_vec, u) and np.array_equal(world.light.cell_v_vec, v)
        assert world.light.cells == cells


def test_sphere_grid_scene_is_deterministic():
    w1, _, _ = scenes.sphere_grid(64, 64)
    w2, _, _ = scenes.sphere_grid(64, 64)
    assert len(w1.objects) == 64
    assert [o.material.color for o in w1.objects] == [o.material.color for o in w2.objects]
    assert sum(1 for o in w1.objects if o.material.reflective > 0) == 32


def test_cosf_restatement_equals_libm_cosf():
    """f32::cos for Sine2D (pattern/sine_2d.rs:40): every reduction path of the glibc routine -- tiny, |x| < pi/4,
    |x| < 120, the large-argument table -- plus random bit patterns; inf / NaN give NaN."""
    libm = C.CDLL("libm.so.6")
    libm.cosf.restype = C.c_float
    libm.cosf.argtypes = [C.c_float]
    rng = np.random.default_rng(99)
    xs = np.concatenate([
        rng.uniform(-1, 1, 60_000), rng.uniform(-130, 130, 120_000), rng.uniform(-1e6, 1e6, 60_000),
        rng.standard_normal(60_000) * 10.0 ** rng.uniform(-20, 38, 60_000),
        [0.0, -0.0, np.pi, np.pi / 2, np.pi / 4, 120.0, 119.99999, 1e-4, 2.0 ** -12, 3.4e38, -3.4e38, 1e-40],
    ]).astype(f32)
    xs = np.concatenate([xs, rng.integers(0, 2 ** 32, 100_000, dtype=np.uint64).astype(np.uint32).view(f32)])
    got = P.cosf_host(xs)
    exp = np.array([libm.cosf(float(v)) for v in xs], dtype=f32)
    same = (got.view(np.uint32) == exp.view(np.uint32)) | (np.isnan(got) & np.isnan(exp))
    assert same.all(), (xs[~same][:5], got[~same][:5], exp[~same][:5])
    assert np.isnan(P.cosf_host(np.array([np.inf, -np.inf, np.nan], dtype=f32))).all()


def test_pattern_and_cone_boundary_errors_without_gpu():
    pat = L.rtc_pattern()
    a = (C.c_float * 3)(1, 1, 1)
    assert L.lib().rtc_pattern_init(C.byref(pat), 0, a, a, None) == L.RTC_ERR_UNSUPPORTED      # NONE is not a pattern
    assert L.lib().rtc_pattern_init(C.byref(pat), 6, a, a, None) == L.RTC_ERR_UNSUPPORTED
    assert L.lib().rtc_pattern_init(C.byref(pat), L.RTC_PATTERN_RINGS, a, a, None) == L.RTC_OK
    assert list(pat.inv) == list(np.eye(4, dtype=f32).reshape(-1))                               # BasePattern::default
    m = L.rtc_material()
    L.lib().rtc_material_default(C.byref(m))
    assert m.pattern.kind == L.RTC_PATTERN_NONE
    obj = L.rtc_object()
    ident = np.eye(4, dtype=f32).reshape(-1)
    assert L.lib().rtc_object_init(C.byref(obj), L.RTC_CONE, ident.ctypes.data_as(L.FP), None) == L.RTC_OK
    assert obj.min_y == -np.inf and obj.max_y == np.inf and obj.closed == 0                      # cone.rs:33-42
    assert L.lib().rtc_object_init(C.byref(obj), 6, ident.ctypes.data_as(L.FP), None) == L.RTC_ERR_UNSUPPORTED


def _libm_powf():
    libm = C.CDLL("libm.so.6")
    libm.powf.restype = C.c_float
    libm.powf.argtypes = [C.c_float, C.c_float]
    return libm.powf


def test_powf_restatement_equals_libm_powf():
    """~6M samples: the specular domain x in (0, 1+], the shininess values the demos use,
    plus random positive x / random y, subnormals and the special cases."""
    powf = _libm_powf()
    rng = np.random.default_rng(1234)
    xs = np.concatenate([
        rng.random(400_000, dtype=f32),                                   # (0,1)
        (f32(1.0) - rng.random(100_000, dtype=f32) * f32(1e-3)),          # just below 1
        (f32(1.0) + rng.random(20_000, dtype=f32) * f32(1e-5)),           # just above 1
        np.exp(rng.uniform(-80, 80, 100_000)).astype(f32),                # wide range
        np.array([1e-40, 1e-45, 5e-39, 1.0, 0.5, 2.0, np.inf, 0.0], dtype=f32),
    ])
    total = 0
    for y in [10.0, 15.0, 50.0, 200.0, 300.0, 1.0, 0.0, 2.5, 1e-3, 1000.0, -3.0, np.inf]:
        ys = np.full(xs.shape, y, dtype=f32)
        got = P.powf_host(xs, ys)
        # vectorised reference through ctypes is slow; sample 1 in 8 for the big blocks, all of the tail
        idx = np.concatenate([np.arange(0, xs.size - 8, 8), np.arange(xs.size - 8, xs.size)])
        exp = np.array([powf(float(xs[i]), float(y)) for i in idx], dtype=f32)
        g = got[idx]
        same = (g == exp) | (np.isnan(g) & np.isnan(exp))
        assert same.all(), (y, xs[idx][~same][:5], g[~same][:5], exp[~same][:5])
        total += idx.size
    ys = rng.uniform(-50, 400, 200_000).astype(f32)
    xr = rng.random(200_000, dtype=f32)
    got = P.powf_host(xr, ys)
    exp = np.array([powf(float(a), float(b)) for a, b in zip(xr[::4], ys[::4])], dtype=f32)
    assert np.array_equal(got[::4], exp)
    assert total > 700_000


def test_to_ppm_matches_oracle_bytes():
    rng = np.random.default_rng(5)
    # the last three are large enough for the formatter to split the rows over threads (>= 2^16 pixels)
    for (w, h) in [(1, 1), (5, 3), (10, 2), (23, 7), (70, 3), (101, 4), (700, 123), (65536, 1), (3, 30000)]:
        img = rng.uniform(-0.2, 1.3, (h, w, 3)).astype(f32)
        img[0, 0] = [np.nan, np.inf, -np.inf]
        assert P.Canvas(w, h, img).to_ppm() == O.to_ppm(img), (w, h)


def test_partition_rows_cover_the_image_once():
    for h in (1, 63, 64, 65, 400, 4096, 1000):
        for n in (1, 2, 3, 4, 8):
            parts = [L.rtc_partition(64, n, p) for p in range(n)]
            rows = [P.lib().rtc_partition_rows(h, C.byref(q)) for q in parts]
            assert sum(rows) == h, (h, n, rows)
    assert P.lib().rtc_partition_rows(100, None) == 100


def test_boundary_errors():
    w = P.default_world()
    w.light = None
    cs = w._c()
    out = np.zeros(3, dtype=f32)
    o, d = P.point(0, 0, -5).reshape(1, 4), P.vector(0, 0, 1).reshape(1, 4)
    st = P.lib().rtc_color_at(C.byref(cs.scene), o.ctypes.data_as(L.FP), d.ctypes.data_as(L.FP), 1, 1, 0,
                              out.ctypes.data_as(L.FP))
    # without a GPU the device check comes first; with one, the missing light is reported (world.rs:66)
    assert st in (L.RTC_ERR_NO_LIGHT, L.RTC_ERR_NO_DEVICE)
    with pytest.raises(P.RtcError) as e:
        P.RectangleLight(P.color(1, 1, 1), P.point(0, 0, 0), P.vector(1, 0, 0), 2, P.vector(0, 1, 0), 2,
                         jitter=lambda: 0.5)._c()
    assert e.value.status == L.RTC_ERR_UNSUPPORTED
    bad = L.rtc_object()
    ident = np.eye(4, dtype=f32).reshape(-1)
    assert P.lib().rtc_object_init(C.byref(bad), 9, ident.ctypes.data_as(L.FP), None) == L.RTC_ERR_UNSUPPORTED
    assert b"shape kind" in P.lib().rtc_last_error()
    cam = L.rtc_camera()
    assert P.lib().rtc_camera_new(0, 10, 1.0, ident.ctypes.data_as(L.FP), C.byref(cam)) == L.RTC_ERR_INVALID_ARG


def test_no_gpu_means_error_not_fallback():
    if P.device_count() > 0:
        pytest.skip("a GPU is visible")
    world, camera, depth = scenes.single_sphere(8, 8)
    with pytest.raises(P.RtcError) as e:
        camera.render(world, depth)
    assert e.value.status == L.RTC_ERR_NO_DEVICE


def test_scene_validation_without_gpu():
    """rtc_scene_validate runs the flattening and its checks on the host: every shipped scene passes, and the
    things the device path refuses are refused here with the same status."""
    import ray_tracer_challenge_amd as P
    for name in ("soft_shadows", "first_scene", "shapes_medley", "patterns_medley", "reflect_refract", "hexagons", "groups_medley",
                 "grouped_grid", "mesh", "first_textures", "skybox", "sphere_grid"):
        world, camera, _ = getattr(scenes, name)(64, 48)
        world.validate(camera)
    world, camera, _ = scenes.hexagons(16, 8)
    cs = world._c()

    def status():
        return L.lib().rtc_scene_validate(C.byref(cs.scene), C.byref(camera._cam))
    assert status() == L.RTC_OK
    cs.groups[1].n_objects = 13          # side 0 sticks out of the hexagon
    assert status() == L.RTC_ERR_INVALID_ARG and b"nest" in L.lib().rtc_last_error()
    cs.groups[1].n_objects = 2
    cs.groups[0].first_object = 2        # not pre-order any more
    assert status() == L.RTC_ERR_INVALID_ARG
    cs.groups[0].first_object = 1
    cs.groups[6].first_object = 40       # outside the object list
    assert status() == L.RTC_ERR_INVALID_ARG
    cs.groups[6].first_object = 11
    assert status() == L.RTC_OK
    proj = np.eye(4, dtype=f32)
    proj[3, 2] = 0.5
    light = P.PointLight(P.point(0, 0, -5), P.color(1, 1, 1))
    for bad, code in ((P.World([P.Sphere(proj)], light), L.RTC_ERR_UNSUPPORTED),
                      (P.World([P.Sphere(None, P.Material(pattern=P.Stripes(transform=proj)))], light), L.RTC_ERR_UNSUPPORTED),
                      (P.World([P.Sphere()], None), L.RTC_ERR_NO_LIGHT),
                      (P.World([P.Sphere(None, P.Material(pattern=P.Pattern(L.RTC_PATTERN_CUBE_MAP, uv=[P.UVCheckers()] * 5)))], light),
                       L.RTC_ERR_UNSUPPORTED)):
        with pytest.raises(P.RtcError) as e:
            bad.validate()
        assert e.value.status == code


def test_the_shipped_library_holds_no_development_switch():
    """RTC_AMD_JIT_SOURCE / _JIT_FLAGS (an environment variable substituting the kernel source or compiler
    flags of a host process) and the other development switches are compiled only into librtc_amd_dev.so; the library that
    ships does not even contain their names.  The policy switches it does read are read when a context is created."""
    import re
    from ray_tracer_challenge_amd import _lib as L
    from tests.conftest import DEV_ONLY_SWITCHES
    assert P.lib().rtc_dev_switches() in (0, 1)
    shipped = open(os.path.join(os.path.dirname(L.DEV_LIB_PATH), "librtc_amd.so"), "rb").read()
    names = set(m.decode() for m in re.findall(rb"RTC_AMD_[A-Z0-9_]+", shipped))
    assert not names & set(DEV_ONLY_SWITCHES), names & set(DEV_ONLY_SWITCHES)
    assert {"RTC_AMD_SPECIALIZE", "RTC_AMD_LIGHT_CULL", "RTC_AMD_JIT_CACHE"} <= names
    if os.path.exists(L.DEV_LIB_PATH):
        dev = open(L.DEV_LIB_PATH, "rb").read()
        dev_names = set(m.decode() for m in re.findall(rb"RTC_AMD_[A-Z0-9_]+", dev))
        assert set(DEV_ONLY_SWITCHES) <= dev_names
        assert L.load(L.DEV_LIB_PATH).rtc_dev_switches() == 1


def test_depth_domain_constants():
    from ray_tracer_challenge_amd import _lib as L
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rtc.h")).read()
    assert "#define RTC_MAX_DEPTH %d" % L.RTC_MAX_DEPTH in hdr and "#define RTC_STACK_DEPTH_BASE %d" % L.RTC_STACK_DEPTH_BASE in hdr
    assert L.RTC_MAX_DEPTH >= 32  # camera.rs:76 takes any i16; the author's own deepest render is 20 (todo.md)
