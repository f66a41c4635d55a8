"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  It mirrors the reference's public API names
(World / Sphere / Plane / Cube / Cylinder / Material / PointLight /
RectangleLight / Camera, translation()/scaling()/... , view_transform()) so the
oracle tests read like the reference's own #[test] functions.

The arithmetic lives in oracle/rtc_oracle.cpp; this file only marshals.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librtc_oracle.so")

SPHERE, PLANE, CUBE, CYLINDER, CONE, TRIANGLE, SMOOTH_TRIANGLE = 0, 1, 2, 3, 4, 5, 6
TEST_SHAPE = 100
PATTERN_NONE, PATTERN_STRIPES, PATTERN_GRADIENT, PATTERN_RINGS, PATTERN_CHECKERS, PATTERN_SINE2D = 0, 1, 2, 3, 4, 5
PATTERN_TEXTURE_MAP, PATTERN_CUBE_MAP = 6, 7
PATTERN_TEST = 100
UV_CHECKERS, UV_ALIGN_CHECK, UV_IMAGE = 1, 2, 3
MAP_SPHERICAL, MAP_PLANAR, MAP_CYLINDRICAL = 1, 2, 3
LIGHT_POINT, LIGHT_RECT = 0, 1
JITTER_CONSTANT, JITTER_CYCLE, JITTER_HASHED = 0, 1, 2

f32 = np.float32
_FP = C.POINTER(C.c_float)


def build(force=False):
    """Compile librtc_oracle.so with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "rtc_oracle.cpp")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src),
                                                   os.path.getmtime(os.path.join(_HERE, "rtc_oracle.h")))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librtc_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _UVPattern(C.Structure):
    _fields_ = [("kind", C.c_int32), ("width", C.c_float), ("height", C.c_float), ("colors", (C.c_float * 3) * 5),
                ("image_width", C.c_uint32), ("image_height", C.c_uint32), ("image_rgb", _FP)]


class _Pattern(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_float * 3), ("b", C.c_float * 3), ("transform", C.c_float * 16),
                ("uv_mapping", C.c_int32), ("n_uv", C.c_int32), ("uv", C.POINTER(_UVPattern))]


class _Material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("ambient", C.c_float), ("diffuse", C.c_float),
                ("specular", C.c_float), ("shininess", C.c_float), ("reflective", C.c_float),
                ("transparency", C.c_float), ("refractive_index", C.c_float), ("pattern", _Pattern)]


class _Shape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("casts_shadow", C.c_int32), ("closed", C.c_int32),
                ("min_y", C.c_float), ("max_y", C.c_float), ("transform", C.c_float * 16),
                ("material", _Material), ("p1", C.c_float * 4), ("p2", C.c_float * 4), ("p3", C.c_float * 4),
                ("n1", C.c_float * 4), ("n2", C.c_float * 4), ("n3", C.c_float * 4)]


class _Light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("intensity", C.c_float * 3), ("position", C.c_float * 4),
                ("corner", C.c_float * 4), ("u_vec", C.c_float * 4), ("u_steps", C.c_int32),
                ("v_vec", C.c_float * 4), ("v_steps", C.c_int32), ("jitter_mode", C.c_int32),
                ("jitter_const", C.c_float), ("jitter_seed", C.c_uint32),
                ("jitter_seq", _FP), ("jitter_seq_len", C.c_int32)]


class _Camera(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("field_of_view", C.c_float),
                ("half_width", C.c_float), ("half_height", C.c_float), ("pixel_size", C.c_float),
                ("transform_inverse", C.c_float * 16)]


class _Comps(C.Structure):
    _fields_ = [("distance", C.c_float), ("object", C.c_int32), ("point", C.c_float * 4),
                ("eye", C.c_float * 4), ("reflectv", C.c_float * 4), ("normal", C.c_float * 4),
                ("over_point", C.c_float * 4), ("under_point", C.c_float * 4), ("inside", C.c_int32),
                ("n1", C.c_float), ("n2", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.rtco_magnitude.restype = C.c_float
        L.rtco_dot.restype = C.c_float
        L.rtco_mat_determinant.restype = C.c_float
        L.rtco_mat_minor.restype = C.c_float
        L.rtco_mat_cofactor.restype = C.c_float
        L.rtco_world_new.restype = C.c_void_p
        L.rtco_world_ray_count.restype = C.c_uint64
        L.rtco_intensity_at.restype = C.c_float
        L.rtco_schlick.restype = C.c_float
        L.rtco_render.restype = C.c_uint64
        L.rtco_render_rows.restype = C.c_uint64
        L.rtco_scale_color.restype = C.c_uint8
        L.rtco_world_new_nodes.restype = C.c_void_p
        L.rtco_node_shininess.restype = C.c_float
        L.rtco_node_set_transformation.argtypes = [C.c_int, _FP]
        L.rtco_node_transformation.argtypes = [C.c_int, _FP]
        L.rtco_node_divide.argtypes = [C.c_int, C.c_uint32]
        L.rtco_to_ppm.restype = C.c_void_p
        L.rtco_jitter_hash.restype = C.c_uint32
        L.rtco_jitter_value.restype = C.c_float
        for name in ("rtco_translation", "rtco_scaling"):
            getattr(L, name).argtypes = [C.c_float] * 3 + [_FP]
        for name in ("rtco_rotation_x", "rtco_rotation_y", "rtco_rotation_z"):
            getattr(L, name).argtypes = [C.c_float, _FP]
        L.rtco_shearing.argtypes = [C.c_float] * 6 + [_FP]
        L.rtco_camera_new.argtypes = [C.c_uint32, C.c_uint32, C.c_float, _FP, C.POINTER(_Camera)]
        L.rtco_position.argtypes = [_FP, _FP, C.c_float, _FP]
        L.rtco_phong.argtypes = [C.c_void_p, C.POINTER(_Material), _FP, _FP, _FP, C.c_float, _FP]
        L.rtco_phong_on.argtypes = [C.c_void_p, C.POINTER(_Shape), _FP, _FP, _FP, C.c_float, _FP]
        L.rtco_scale_color.argtypes = [C.c_float]
        L.rtco_uv_color_at.argtypes = [C.POINTER(_UVPattern), C.c_float, C.c_float, _FP]
        L.rtco_canvas_from_ppm.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_void_p)]
        L.rtco_normal_at_uv.argtypes = [C.POINTER(_Shape), _FP, C.c_float, C.c_float, _FP]
        L.rtco_jitter_value.argtypes = [C.c_uint32]
        _lib = L
    return _lib


# ----------------------------------------------------------------- helpers
def _a(x, n=None):
    arr = np.ascontiguousarray(np.asarray(x, dtype=f32).reshape(-1))
    if n is not None:
        assert arr.size == n, (arr.size, n)
    return arr


def _p(arr):
    return arr.ctypes.data_as(_FP)


def point(x, y, z):
    return np.array([x, y, z, 1.0], dtype=f32)


def vector(x, y, z):
    return np.array([x, y, z, 0.0], dtype=f32)


def color(r, g, b):
    return np.array([r, g, b], dtype=f32)


def identity_4x4():
    return np.eye(4, dtype=f32)


def _mat_out(fn, *args):
    out = np.zeros(16, dtype=f32)
    fn(*args, _p(out))
    return out.reshape(4, 4)


def translation(x, y, z):
    return _mat_out(lib().rtco_translation, f32(x), f32(y), f32(z))


def scaling(x, y, z):
    return _mat_out(lib().rtco_scaling, f32(x), f32(y), f32(z))


def rotation_x(r):
    return _mat_out(lib().rtco_rotation_x, f32(r))


def rotation_y(r):
    return _mat_out(lib().rtco_rotation_y, f32(r))


def rotation_z(r):
    return _mat_out(lib().rtco_rotation_z, f32(r))


def shearing(xy, xz, yx, yz, zx, zy):
    return _mat_out(lib().rtco_shearing, *[f32(v) for v in (xy, xz, yx, yz, zx, zy)])


def view_transform(frm, to, up):
    a, b, c = _a(frm, 4), _a(to, 4), _a(up, 4)
    out = np.zeros(16, dtype=f32)
    lib().rtco_view_transform(_p(a), _p(b), _p(c), _p(out))
    return out.reshape(4, 4)


def mat_mul(a, b):
    x, y = _a(a, 16), _a(b, 16)
    out = np.zeros(16, dtype=f32)
    lib().rtco_mat_mul(_p(x), _p(y), _p(out))
    return out.reshape(4, 4)


def chain(*ms):
    """a * b * c with the reference's left-to-right operator evaluation."""
    out = ms[0]
    for m in ms[1:]:
        out = mat_mul(out, m)
    return out


def mat_vec(a, v):
    x, y = _a(a, 16), _a(v, 4)
    out = np.zeros(4, dtype=f32)
    lib().rtco_mat_vec(_p(x), _p(y), _p(out))
    return out


def transpose(a):
    a = np.asarray(a, dtype=f32)
    n = a.shape[0]
    x = _a(a)
    out = np.zeros(n * n, dtype=f32)
    lib().rtco_mat_transpose(_p(x), n, _p(out))
    return out.reshape(n, n)


def determinant(a):
    a = np.asarray(a, dtype=f32)
    x = _a(a)
    return f32(lib().rtco_mat_determinant(_p(x), a.shape[0]))


def submatrix(a, r, c):
    a = np.asarray(a, dtype=f32)
    n = a.shape[0]
    x = _a(a)
    out = np.zeros((n - 1) * (n - 1), dtype=f32)
    lib().rtco_mat_submatrix(_p(x), n, r, c, _p(out))
    return out.reshape(n - 1, n - 1)


def minor(a, r, c):
    a = np.asarray(a, dtype=f32)
    x = _a(a)
    return f32(lib().rtco_mat_minor(_p(x), a.shape[0], r, c))


def cofactor(a, r, c):
    a = np.asarray(a, dtype=f32)
    x = _a(a)
    return f32(lib().rtco_mat_cofactor(_p(x), a.shape[0], r, c))


def inverse(a):
    a = np.asarray(a, dtype=f32)
    n = a.shape[0]
    x = _a(a)
    out = np.zeros(n * n, dtype=f32)
    lib().rtco_mat_inverse(_p(x), n, _p(out))
    return out.reshape(n, n)


def magnitude(v):
    x = _a(v, 4)
    return f32(lib().rtco_magnitude(_p(x)))


def norm(v):
    x = _a(v, 4)
    out = np.zeros(4, dtype=f32)
    lib().rtco_norm(_p(x), _p(out))
    return out


def dot(a, b):
    x, y = _a(a, 4), _a(b, 4)
    return f32(lib().rtco_dot(_p(x), _p(y)))


def cross(a, b):
    x, y = _a(a, 4), _a(b, 4)
    out = np.zeros(4, dtype=f32)
    lib().rtco_cross(_p(x), _p(y), _p(out))
    return out


def reflect(v, n):
    x, y = _a(v, 4), _a(n, 4)
    out = np.zeros(4, dtype=f32)
    lib().rtco_reflect(_p(x), _p(y), _p(out))
    return out


def position(o, d, t):
    x, y = _a(o, 4), _a(d, 4)
    out = np.zeros(4, dtype=f32)
    lib().rtco_position(_p(x), _p(y), f32(t), _p(out))
    return out


def hit(ts):
    """Intersection::hit over a list of distances; index or None."""
    x = _a(ts)
    i = lib().rtco_hit(_p(x), x.size)
    return None if i < 0 else i


def aabb_intersection(o, d, mn, mx):
    a, b, c, e = _a(o, 4), _a(d, 4), _a(mn, 4), _a(mx, 4)
    out = np.zeros(2, dtype=f32)
    ok = lib().rtco_aabb_intersection(_p(a), _p(b), _p(c), _p(e), _p(out))
    return (out[0], out[1]) if ok else None


def scale_color(c):
    return int(lib().rtco_scale_color(f32(c)))


def quantize(rgb):
    x = np.ascontiguousarray(rgb, dtype=f32)
    out = np.zeros(x.shape, dtype=np.uint8)
    lib().rtco_quantize(_p(x), C.c_uint64(x.size), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def to_ppm(rgb):
    """Canvas::to_ppm for an (h, w, 3) float32 image -> bytes."""
    x = np.ascontiguousarray(rgb, dtype=f32)
    h, w, _ = x.shape
    n = C.c_uint64(0)
    ptr = lib().rtco_to_ppm(_p(x), w, h, C.byref(n))
    try:
        return C.string_at(ptr, n.value)
    finally:
        lib().rtco_free(C.c_void_p(ptr))


def jitter_hash(seed, pixel, path, cell, draw):
    return int(lib().rtco_jitter_hash(C.c_uint32(seed), C.c_uint32(pixel), C.c_uint32(path),
                                      C.c_uint32(cell), C.c_uint32(draw)))


def jitter_value(h):
    return f32(lib().rtco_jitter_value(C.c_uint32(h)))


# ------------------------------------------------------------ scene classes
class Pattern:
    """pattern/*.rs: two colours and a pattern->object transform (set_transformation inverts it)."""

    def __init__(self, kind, a=(1, 1, 1), b=(0, 0, 0), transform=None, uv_mapping=0, uv=()):
        self.kind = kind
        self.a, self.b = tuple(float(c) for c in a), tuple(float(c) for c in b)
        self.transform = identity_4x4() if transform is None else np.asarray(transform, dtype=f32)
        self.uv_mapping, self.uv = uv_mapping, list(uv)

    def set_transformation(self, t):
        self.transform = np.asarray(t, dtype=f32)

    def _uv_array(self, struct_type):
        """The C array of this pattern's UV patterns.  The C struct returned by _c() -- and every copy of it, e.g. one per
        leaf of a mesh whose triangles share this Pattern through Material.copy() -- BORROWS it, so it must outlive them
        all: it is built once per state of the UV patterns and kept on the Pattern (replacing it on every _c() call freed
        the array under the earlier copies)."""
        key = tuple((type(u).__name__, tuple(sorted((k, id(v) if hasattr(v, "__dict__") or hasattr(v, "shape") else v)
                                                      for k, v in vars(u).items() if not k.startswith("_")))) for u in self.uv)
        cached = getattr(self, "_uv_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, (struct_type * len(self.uv))(*[u._c() for u in self.uv]), list(self.uv))
            self._uv_cache = cached
        return cached[1]

    def _c(self):
        p = _Pattern()
        p.kind = self.kind
        p.a[:] = [f32(c) for c in self.a]
        p.b[:] = [f32(c) for c in self.b]
        p.transform[:] = [f32(v) for v in self.transform.reshape(-1)]
        if self.uv:
            arr = self._uv_array(_UVPattern)
            p.uv_mapping, p.n_uv, p.uv = self.uv_mapping, len(self.uv), arr
        return p

    def color_at_world(self, point):
        c, a = self._c(), _a(point, 4)
        out = np.zeros(3, dtype=f32)
        lib().rtco_pattern_color_at_world(C.byref(c), _p(a), _p(out))
        return out

    def color_at_object(self, world_point, shape):
        c, s, a = self._c(), shape._c(), _a(world_point, 4)
        out = np.zeros(3, dtype=f32)
        lib().rtco_pattern_color_at_object(C.byref(c), C.byref(s), _p(a), _p(out))
        return out


def Stripes(a=(1, 1, 1), b=(0, 0, 0), transform=None):
    return Pattern(PATTERN_STRIPES, a, b, transform)


def Gradient(a=(1, 1, 1), b=(0, 0, 0), transform=None):
    return Pattern(PATTERN_GRADIENT, a, b, transform)


def Rings(a=(1, 1, 1), b=(0, 0, 0), transform=None):
    return Pattern(PATTERN_RINGS, a, b, transform)


def Checkers(a=(1, 1, 1), b=(0, 0, 0), transform=None):
    return Pattern(PATTERN_CHECKERS, a, b, transform)


def Sine2D(a=(1, 1, 1), b=(0, 0, 0), transform=None):
    return Pattern(PATTERN_SINE2D, a, b, transform)


# ---- pattern/uv.rs ----
class UVPatternBase:
    def color_at(self, u, v):
        c = self._c()
        out = np.zeros(3, dtype=f32)
        lib().rtco_uv_color_at(C.byref(c), f32(u), f32(v), _p(out))
        return out


class UVCheckers(UVPatternBase):
    def __init__(self, width=1.0, height=1.0, a=(1, 1, 1), b=(0, 0, 0)):
        self.width, self.height, self.a, self.b = width, height, tuple(a), tuple(b)

    def _c(self):
        u = _UVPattern()
        u.kind, u.width, u.height = UV_CHECKERS, f32(self.width), f32(self.height)
        u.colors[0][:] = [f32(c) for c in self.a]
        u.colors[1][:] = [f32(c) for c in self.b]
        return u


class AlignCheck(UVPatternBase):
    def __init__(self, main=(1, 1, 1), ul=(1, 0, 0), ur=(1, 1, 0), bl=(0, 1, 0), br=(0, 1, 1)):
        self.colors = [tuple(c) for c in (main, ul, ur, bl, br)]

    def _c(self):
        u = _UVPattern()
        u.kind = UV_ALIGN_CHECK
        for k, col in enumerate(self.colors):
            u.colors[k][:] = [f32(c) for c in col]
        return u


class UVImage(UVPatternBase):
    def __init__(self, canvas):
        """canvas: (h, w, 3) f32 array, e.g. from canvas_from_ppm()"""
        self.canvas = np.ascontiguousarray(canvas, dtype=f32)

    def _c(self):
        u = _UVPattern()
        u.kind = UV_IMAGE
        u.image_height, u.image_width = self.canvas.shape[0], self.canvas.shape[1]
        u.image_rgb = _p(self.canvas)
        return u


class SphericalMap:
    kind = MAP_SPHERICAL


class PlanarMap:
    kind = MAP_PLANAR


class CylindricalMap:
    kind = MAP_CYLINDRICAL


def point_to_uv(mapping, p):
    a, out = _a(p, 4), np.zeros(2, dtype=f32)
    lib().rtco_point_to_uv(int(mapping.kind), _p(a), _p(out))
    return out[0], out[1]


FACES = ("front", "back", "left", "right", "up", "down")


def face_from_point(p):
    a = _a(p, 4)
    return FACES[lib().rtco_face_from_point(_p(a))]


def cube_uv(face, p):
    a, out = _a(p, 4), np.zeros(2, dtype=f32)
    lib().rtco_cube_uv(FACES.index(face), _p(a), _p(out))
    return out[0], out[1]


def TextureMap(uv_pattern, uv_mapping, transform=None):
    """TextureMap::new(uv_pattern, uv_mapping) -- pattern/uv.rs:68-76"""
    return Pattern(PATTERN_TEXTURE_MAP, transform=transform, uv_mapping=uv_mapping.kind, uv=[uv_pattern])


def CubicMap(front, back, left, right, up, down, transform=None):
    """CubicMap::new -- pattern/uv.rs:207-231"""
    return Pattern(PATTERN_CUBE_MAP, transform=transform, uv=[front, back, left, right, up, down])


class PpmParseError(ValueError):
    VARIANTS = ("IoError", "IncorrectFormat", "ParseIntError", "MalformedDimensionHeader", "PixelOutOfBounds (panic)")

    def __init__(self, code):
        super().__init__(self.VARIANTS[code - 1])
        self.kind = self.VARIANTS[code - 1]


def canvas_from_ppm(text):
    """canvas.rs:120-197 -> (h, w, 3) f32"""
    if isinstance(text, str):
        text = text.encode()
    w, h, rgb = C.c_uint32(), C.c_uint32(), C.c_void_p()
    rc = lib().rtco_canvas_from_ppm(text, len(text), C.byref(w), C.byref(h), C.byref(rgb))
    if rc:
        raise PpmParseError(rc)
    try:
        n = w.value * h.value * 3
        arr = np.ctypeslib.as_array(C.cast(rgb, _FP), shape=(max(n, 1),))[:n].copy()
    finally:
        lib().rtco_free(rgb)
    return arr.reshape(h.value, w.value, 3)


def TestPattern(transform=None):
    """pattern.rs:66-89: returns the pattern-space point as the colour."""
    return Pattern(PATTERN_TEST, transform=transform)


TestPattern.__test__ = False


class Material:
    """material.rs:18-51 defaults."""

    def __init__(self, color=(1, 1, 1), ambient=0.1, diffuse=0.9, specular=0.9, shininess=200.0,
                 reflective=0.0, transparency=0.0, refractive_index=1.0, pattern=None):
        self.color = tuple(float(c) for c in color)
        self.ambient, self.diffuse, self.specular = ambient, diffuse, specular
        self.shininess, self.reflective = shininess, reflective
        self.transparency, self.refractive_index = transparency, refractive_index
        self.pattern = pattern

    def copy(self, **kw):
        m = Material(self.color, self.ambient, self.diffuse, self.specular, self.shininess,
                     self.reflective, self.transparency, self.refractive_index, self.pattern)
        for k, v in kw.items():
            setattr(m, k, v)
        return m

    def _c(self):
        m = _Material()
        m.color[:] = [f32(c) for c in self.color]
        for k in ("ambient", "diffuse", "specular", "shininess", "reflective", "transparency",
                  "refractive_index"):
            setattr(m, k, f32(getattr(self, k)))
        if self.pattern is not None:
            m.pattern = self.pattern._c()
        return m


class Shape:
    def __init__(self, kind, transform=None, material=None, casts_shadow=True,
                 minimum_y=-np.inf, maximum_y=np.inf, closed=False):
        self.kind = kind
        self.transform = identity_4x4() if transform is None else np.asarray(transform, dtype=f32)
        self.material = Material() if material is None else material
        self.casts_shadow = casts_shadow
        self.minimum_y, self.maximum_y, self.closed = minimum_y, maximum_y, closed
        self.points, self.normals = None, None  # triangles

    def set_transformation(self, t):
        self.transform = np.asarray(t, dtype=f32)

    def set_material(self, m):
        self.material = m

    def _c(self):
        s = _Shape()
        s.kind = self.kind
        s.casts_shadow = int(self.casts_shadow)
        s.closed = int(self.closed)
        s.min_y = f32(self.minimum_y)
        s.max_y = f32(self.maximum_y)
        s.transform[:] = [f32(v) for v in self.transform.reshape(-1)]
        s.material = self.material._c()
        if self.points is not None:
            for name, p in zip(("p1", "p2", "p3"), self.points):
                getattr(s, name)[:] = [f32(v) for v in p]
        if self.normals is not None:
            for name, n in zip(("n1", "n2", "n3"), self.normals):
                getattr(s, name)[:] = [f32(v) for v in n]
        return s

    # triangle.rs / smooth_triangle.rs
    def local_intersect_uv(self, o, d):
        s, a, b = self._c(), _a(o, 4), _a(d, 4)
        ts, us, vs = np.zeros(4, dtype=f32), np.zeros(4, dtype=f32), np.zeros(4, dtype=f32)
        n = lib().rtco_local_intersect_uv(C.byref(s), _p(a), _p(b), _p(ts), _p(us), _p(vs))
        return [(ts[i], us[i], vs[i]) for i in range(n)]

    def normal_at_uv(self, p, u, v):
        s, a = self._c(), _a(p, 4)
        out = np.zeros(4, dtype=f32)
        lib().rtco_normal_at_uv(C.byref(s), _p(a), f32(u), f32(v), _p(out))
        return out

    def triangle_fields(self):
        s = self._c()
        out = [np.zeros(4, dtype=f32) for _ in range(3)]
        lib().rtco_triangle_fields(C.byref(s), *[_p(x) for x in out])
        return out

    # Shape::local_intersect / intersect / local_norm_at / normal_at
    def local_intersect(self, o, d):
        s, a, b = self._c(), _a(o, 4), _a(d, 4)
        ts = np.zeros(4, dtype=f32)
        n = lib().rtco_local_intersect(C.byref(s), _p(a), _p(b), _p(ts))
        return [ts[i] for i in range(n)]

    def intersect(self, o, d):
        s, a, b = self._c(), _a(o, 4), _a(d, 4)
        ts, oo, od = np.zeros(4, dtype=f32), np.zeros(4, dtype=f32), np.zeros(4, dtype=f32)
        n = lib().rtco_shape_intersect(C.byref(s), _p(a), _p(b), _p(ts), _p(oo), _p(od))
        return [ts[i] for i in range(n)], oo, od

    def local_norm_at(self, p):
        s, a = self._c(), _a(p, 4)
        out = np.zeros(4, dtype=f32)
        lib().rtco_local_normal_at(C.byref(s), _p(a), _p(out))
        return out

    def normal_at(self, p):
        s, a = self._c(), _a(p, 4)
        out = np.zeros(4, dtype=f32)
        lib().rtco_normal_at(C.byref(s), _p(a), _p(out))
        return out


def Sphere(transform=None, material=None, **kw):
    return Shape(SPHERE, transform, material, **kw)


def Plane(transform=None, material=None, **kw):
    return Shape(PLANE, transform, material, **kw)


def Cube(transform=None, material=None, **kw):
    return Shape(CUBE, transform, material, **kw)


def Cylinder(transform=None, material=None, **kw):
    return Shape(CYLINDER, transform, material, **kw)


def Cone(transform=None, material=None, **kw):
    return Shape(CONE, transform, material, **kw)


def Triangle(p1, p2, p3, transform=None, material=None, **kw):
    """Triangle::new(p1, p2, p3) -- shape/triangle.rs:19-33"""
    t = Shape(TRIANGLE, transform, material, **kw)
    t.points = [_a(p, 4) for p in (p1, p2, p3)]
    return t


def SmoothTriangle(p1, p2, p3, n1, n2, n3, transform=None, material=None, **kw):
    """SmoothTriangle::new -- shape/smooth_triangle.rs:17-26"""
    t = Shape(SMOOTH_TRIANGLE, transform, material, **kw)
    t.points = [_a(p, 4) for p in (p1, p2, p3)]
    t.normals = [_a(n, 4) for n in (n1, n2, n3)]
    return t


def TestShape(transform=None, material=None, **kw):
    """shape/test_shape.rs: never intersects; local normal = (2x, 3y, 4z)."""
    return Shape(TEST_SHAPE, transform, material, **kw)


TestShape.__test__ = False


# ---------------------------------------------------------------- bounding_box.rs
class BoundingBox:
    """bounding_box.rs:7-11; min / max are points (w = 1)."""

    def __init__(self, mn, mx):
        self.min, self.max = _a(mn, 4).copy(), _a(mx, 4).copy()

    @staticmethod
    def empty():
        mn, mx = np.zeros(4, dtype=f32), np.zeros(4, dtype=f32)
        lib().rtco_bbox_empty(_p(mn), _p(mx))
        return BoundingBox(mn, mx)

    with_bounds = staticmethod(lambda mn, mx: BoundingBox(mn, mx))

    def add_point(self, p):
        a = _a(p, 4)
        lib().rtco_bbox_add_point(_p(self.min), _p(self.max), _p(a))

    def add_bounding_box(self, other):
        lib().rtco_bbox_add(_p(self.min), _p(self.max), _p(other.min), _p(other.max))

    def contains_point(self, p):
        a = _a(p, 4)
        return bool(lib().rtco_bbox_contains_point(_p(self.min), _p(self.max), _p(a)))

    def contains_bounding_box(self, other):
        return bool(lib().rtco_bbox_contains(_p(self.min), _p(self.max), _p(other.min), _p(other.max)))

    def transform(self, m):
        mm, mn, mx = _a(m, 16), np.zeros(4, dtype=f32), np.zeros(4, dtype=f32)
        lib().rtco_bbox_transform(_p(self.min), _p(self.max), _p(mm), _p(mn), _p(mx))
        return BoundingBox(mn, mx)

    def intersects(self, o, d):
        return aabb_intersection(o, d, self.min, self.max) is not None

    def split(self):
        out = [np.zeros(4, dtype=f32) for _ in range(4)]
        lib().rtco_bbox_split(_p(self.min), _p(self.max), *[_p(x) for x in out])
        return BoundingBox(out[0], out[1]), BoundingBox(out[2], out[3])


def _shape_bounds(shape, parent_space):
    s, mn, mx = shape._c(), np.zeros(4, dtype=f32), np.zeros(4, dtype=f32)
    lib().rtco_shape_bounding_box(C.byref(s), int(parent_space), _p(mn), _p(mx))
    return BoundingBox(mn, mx)


Shape.bounding_box = lambda self: _shape_bounds(self, False)
Shape.parent_space_bounding_box = lambda self: _shape_bounds(self, True)


# ---------------------------------------------------------------- shape/group.rs
def _node_of(x):
    """Arena node of a shape handed to a group (the Box<dyn Shape> moves into the tree)."""
    if isinstance(x, (GroupShape, NodeRef)):
        return x.node
    if getattr(x, "_node", None) is None:
        c = x._c()
        x._node = int(lib().rtco_node_shape(C.byref(c)))
    return x._node


class NodeRef:
    """A leaf shape living inside a group tree (what get_children() hands back)."""

    is_group = False

    def __init__(self, node):
        self.node = node

    def get_unique_id(self):
        return self.node

    def transformation(self):
        out = np.zeros(16, dtype=f32)
        lib().rtco_node_transformation(self.node, _p(out))
        return out.reshape(4, 4)

    @property
    def shininess(self):
        return f32(lib().rtco_node_shininess(self.node))

    def world_to_object_point(self, p):
        a, out = _a(p, 4), np.zeros(4, dtype=f32)
        lib().rtco_node_world_to_object(self.node, _p(a), _p(out))
        return out

    def normal_at(self, p):
        a, out = _a(p, 4), np.zeros(4, dtype=f32)
        lib().rtco_node_normal_at(self.node, _p(a), _p(out))
        return out

    def _bbox(self, fn):
        mn, mx = np.zeros(4, dtype=f32), np.zeros(4, dtype=f32)
        fn(self.node, _p(mn), _p(mx))
        return BoundingBox(mn, mx)

    def bounding_box(self):
        return self._bbox(lib().rtco_node_bounding_box)

    def parent_space_bounding_box(self):
        return self._bbox(lib().rtco_node_parent_space_bounding_box)

    def set_transformation(self, t):
        m = _a(t, 16)
        lib().rtco_node_set_transformation(self.node, _p(m))

    def set_material(self, material):
        m = material._c()
        lib().rtco_node_set_material(self.node, C.byref(m))

    def divide(self, threshold):
        lib().rtco_node_divide(self.node, int(threshold))

    def intersect(self, o, d):
        """Shape::intersect: [(distance, leaf NodeRef)] in push order."""
        a, b = _a(o, 4), _a(d, 4)
        cap = 256
        ts, leaves = np.zeros(cap, dtype=f32), (C.c_int * cap)()
        n = lib().rtco_node_intersect(self.node, _p(a), _p(b), _p(ts), leaves, cap)
        return [(ts[i], NodeRef(leaves[i])) for i in range(min(n, cap))]


class GroupShape(NodeRef):
    """shape/group.rs: add_child / set_transformation bake the group's transform into the children."""

    is_group = True

    def __init__(self, node=None):
        super().__init__(int(lib().rtco_node_group()) if node is None else node)

    @staticmethod
    def with_children(children):
        ids = [_node_of(c) for c in children]
        arr = (C.c_int * max(len(ids), 1))(*ids)
        return GroupShape(int(lib().rtco_node_group_with_children(arr, len(ids))))

    def add_child(self, child):
        lib().rtco_node_add_child(self.node, _node_of(child))

    def get_children(self):
        cap = 4096
        arr = (C.c_int * cap)()
        n = lib().rtco_node_children(self.node, arr, cap)
        assert n <= cap
        return [GroupShape(arr[i]) if lib().rtco_node_is_group(arr[i]) else NodeRef(arr[i]) for i in range(n)]

    local_intersect = NodeRef.intersect


class PointLight:
    def __init__(self, position, intensity):
        self.position, self.intensity = _a(position, 4), _a(intensity, 3)


class RectangleLight:
    """jitter: ('constant', c) | ('cycle', [..]) | ('hashed', seed)."""

    def __init__(self, intensity, corner, u_vec, u_steps, v_vec, v_steps, jitter=("constant", 0.5)):
        self.intensity, self.corner = _a(intensity, 3), _a(corner, 4)
        self.u_vec, self.v_vec = _a(u_vec, 4), _a(v_vec, 4)
        self.u_steps, self.v_steps, self.jitter = u_steps, v_steps, jitter


class World:
    def __init__(self, objects=(), light=None):
        self.objects = list(objects)
        self.light = light
        self._h = None
        self._keep = None

    def _handle(self):
        if self._h is None:
            n = len(self.objects)
            arr = (_Shape * max(n, 1))()
            for i, o in enumerate(self.objects):
                if not isinstance(o, NodeRef):
                    arr[i] = o._c()
            l = _Light()
            lt = self.light
            if isinstance(lt, PointLight):
                l.kind = LIGHT_POINT
                l.intensity[:] = list(lt.intensity)
                l.position[:] = list(lt.position)
            elif isinstance(lt, RectangleLight):
                l.kind = LIGHT_RECT
                l.intensity[:] = list(lt.intensity)
                l.corner[:] = list(lt.corner)
                l.u_vec[:] = list(lt.u_vec)
                l.v_vec[:] = list(lt.v_vec)
                l.u_steps, l.v_steps = lt.u_steps, lt.v_steps
                mode, arg = lt.jitter
                if mode == "constant":
                    l.jitter_mode, l.jitter_const = JITTER_CONSTANT, f32(arg)
                elif mode == "cycle":
                    seq = _a(arg)
                    self._keep = seq
                    l.jitter_mode, l.jitter_seq, l.jitter_seq_len = JITTER_CYCLE, _p(seq), seq.size
                else:
                    l.jitter_mode, l.jitter_seed = JITTER_HASHED, int(arg)
            else:
                raise ValueError("World light should be set")  # world.rs:66
            if any(isinstance(o, NodeRef) for o in self.objects):
                roots = [_node_of(o) for o in self.objects]
                self._h = C.c_void_p(lib().rtco_world_new_nodes((C.c_int * len(roots))(*roots), len(roots), C.byref(l)))
            else:
                self._h = C.c_void_p(lib().rtco_world_new(arr, n, C.byref(l)))
        return self._h

    def invalidate(self):
        if self._h is not None:
            lib().rtco_world_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.invalidate()
        except Exception:
            pass

    def set_pixel(self, idx):
        lib().rtco_world_set_pixel(self._handle(), C.c_uint32(idx))

    @property
    def ray_count(self):
        return int(lib().rtco_world_ray_count(self._handle()))

    def shape_inverse(self, i):
        a, b = np.zeros(16, dtype=f32), np.zeros(16, dtype=f32)
        lib().rtco_world_shape_inverse(self._handle(), i, _p(a), _p(b))
        return a.reshape(4, 4), b.reshape(4, 4)

    def light_info(self):
        pos, u, v = np.zeros(4, dtype=f32), np.zeros(4, dtype=f32), np.zeros(4, dtype=f32)
        cells = C.c_int(0)
        lib().rtco_light_info(self._handle(), _p(pos), _p(u), _p(v), C.byref(cells))
        return pos, u, v, cells.value

    def intersect(self, o, d):
        a, b = _a(o, 4), _a(d, 4)
        cap = 4 * max(len(self.objects), 1)
        ts, objs = np.zeros(cap, dtype=f32), np.zeros(cap, dtype=np.int32)
        n = lib().rtco_intersect(self._handle(), _p(a), _p(b), _p(ts), objs.ctypes.data_as(C.POINTER(C.c_int)),
                                 cap)
        return ts[:n].copy(), objs[:n].copy()

    def color_at(self, o, d, depth):
        a, b = _a(o, 4), _a(d, 4)
        out = np.zeros(3, dtype=f32)
        lib().rtco_color_at(self._handle(), _p(a), _p(b), int(depth), _p(out))
        return out

    def is_shadowed(self, light_position, p):
        a, b = _a(light_position, 4), _a(p, 4)
        return bool(lib().rtco_is_shadowed(self._handle(), _p(a), _p(b)))

    def intensity_at(self, p):
        a = _a(p, 4)
        return f32(lib().rtco_intensity_at(self._handle(), _p(a)))

    def point_on_light(self, u, v):
        out = np.zeros(4, dtype=f32)
        lib().rtco_point_on_light(self._handle(), int(u), int(v), _p(out))
        return out

    def precompute_values(self, o, d, hit_index, xs):
        """xs: list of (distance, object_index)."""
        a, b = _a(o, 4), _a(d, 4)
        ts = _a([x[0] for x in xs])
        objs = np.ascontiguousarray([x[1] for x in xs], dtype=np.int32)
        c = _Comps()
        lib().rtco_precompute(self._handle(), _p(a), _p(b), int(hit_index), _p(ts),
                              objs.ctypes.data_as(C.POINTER(C.c_int)), len(xs), C.byref(c))
        return c

    def shade_hit(self, comps, depth):
        out = np.zeros(3, dtype=f32)
        lib().rtco_shade_hit(self._handle(), C.byref(comps), int(depth), _p(out))
        return out

    def reflected_color(self, comps, depth):
        out = np.zeros(3, dtype=f32)
        lib().rtco_reflected_color(self._handle(), C.byref(comps), int(depth), _p(out))
        return out

    def refracted_color(self, comps, depth):
        out = np.zeros(3, dtype=f32)
        lib().rtco_refracted_color(self._handle(), C.byref(comps), int(depth), _p(out))
        return out

    def phong_lighting(self, material, p, eye, n, light_intensity, shape=None):
        """phong_lighting.rs:12-63; `shape` is the lit object (default: the tests' any_shape(), a unit sphere)."""
        a, b, c = _a(p, 4), _a(eye, 4), _a(n, 4)
        out = np.zeros(3, dtype=f32)
        if shape is None:
            m = material._c()
            lib().rtco_phong(self._handle(), C.byref(m), _p(a), _p(b), _p(c), f32(light_intensity), _p(out))
        else:
            s = Shape(shape.kind, shape.transform, material)._c()
            lib().rtco_phong_on(self._handle(), C.byref(s), _p(a), _p(b), _p(c), f32(light_intensity), _p(out))
        return out


def schlick_reflectance(comps):
    return f32(lib().rtco_schlick(C.byref(comps)))


def arr4(c_arr):
    return np.array(list(c_arr), dtype=f32)


def default_world():
    """World::default(), world.rs:32-48."""
    m = Material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2)
    s1 = Sphere(identity_4x4(), m)
    s2 = Sphere(scaling(0.5, 0.5, 0.5), Material())
    return World([s1, s2], PointLight(point(-10.0, 10.0, -10.0), color(1, 1, 1)))


class Camera:
    def __init__(self, width, height, field_of_view, transform):
        t = _a(transform, 16)
        self._c = _Camera()
        lib().rtco_camera_new(int(width), int(height), f32(field_of_view), _p(t), C.byref(self._c))
        self.width, self.height = int(width), int(height)

    @property
    def pixel_size(self):
        return f32(self._c.pixel_size)

    @property
    def half_width(self):
        return f32(self._c.half_width)

    @property
    def half_height(self):
        return f32(self._c.half_height)

    @property
    def transform_inverse(self):
        return np.array(list(self._c.transform_inverse), dtype=f32).reshape(4, 4)

    def ray_for_pixel(self, x, y):
        o, d = np.zeros(4, dtype=f32), np.zeros(4, dtype=f32)
        lib().rtco_ray_for_pixel(C.byref(self._c), int(x), int(y), _p(o), _p(d))
        return o, d

    def render(self, world, depth, threads=1, rows=None):
        """Camera::render -> ((h, w, 3) float32 image, rays traced)."""
        img = np.zeros((self.height, self.width, 3), dtype=f32)
        if rows is None:
            rays = lib().rtco_render(world._handle(), C.byref(self._c), int(depth), int(threads), _p(img))
        else:
            rays = lib().rtco_render_rows(world._handle(), C.byref(self._c), int(depth), int(threads),
                                          int(rows[0]), int(rows[1]), _p(img))
        return img, int(rays)
