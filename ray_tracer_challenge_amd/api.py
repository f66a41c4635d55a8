"""Host-side mirror of the reference's scene API for the render hot path.

Names, argument meaning and defaults follow the Rust crate
(lib/src/{tuple,matrix,transformations,material,world,camera,canvas}.rs and
lib/src/{shape,light}/*.rs) so scenes and tests written against the reference
carry over:  World{objects, light}, Sphere/Plane/Cube/Cylinder::build(transform,
material), Material builder defaults, PointLight::new, RectangleLight::new,
Camera::new, camera.render(world, depth) -> Canvas, canvas.to_ppm().

Everything numeric happens behind the C ABI (include/rtc.h): scene math on the
host in librtc_amd.so, rendering on the MI355X.  There is no fallback path.
"""
import ctypes as C

import numpy as np

from . import _lib as L

f32 = np.float32


def _a(x, n=None):
    arr = np.ascontiguousarray(np.asarray(x, dtype=f32).reshape(-1))
    if n is not None and arr.size != n:
        raise ValueError("expected %d values, got %d" % (n, arr.size))
    return arr


_FLOAT_ARRAYS = {}


def _p(arr):
    """float* for a contiguous float32 array (a ctypes array sharing its buffer: a third of data_as's cost, which
    dominates building a mesh through this API)."""
    n = arr.size
    t = _FLOAT_ARRAYS.get(n)
    if t is None:
        t = _FLOAT_ARRAYS[n] = C.c_float * n
    try:
        return t.from_buffer(arr)
    except (TypeError, ValueError):  # read-only or empty buffers
        return arr.ctypes.data_as(L.FP)


# ------------------------------------------------------------- tuple.rs, color.rs
def point(x, y, z):
    """point!(x, y, z) -- tuple.rs:72-77"""
    return np.array([x, y, z, 1.0], dtype=f32)


def vector(x, y, z):
    """vector!(x, y, z) -- tuple.rs:80-85"""
    return np.array([x, y, z, 0.0], dtype=f32)


def color(r, g, b):
    """color!(r, g, b) -- color.rs:27-31"""
    return np.array([r, g, b], dtype=f32)


def magnitude(v):
    x = _a(v, 4)
    return f32(L.lib().rtc_magnitude(_p(x)))


def norm(v):
    x, out = _a(v, 4), np.zeros(4, dtype=f32)
    L.lib().rtc_norm(_p(x), _p(out))
    return out


def dot(a, b):
    x, y = _a(a, 4), _a(b, 4)
    return f32(L.lib().rtc_dot(_p(x), _p(y)))


def cross(a, b):
    x, y, out = _a(a, 4), _a(b, 4), np.zeros(4, dtype=f32)
    L.lib().rtc_cross(_p(x), _p(y), _p(out))
    return out


def reflect(in_vector, normal_vector):
    """Ray::reflect -- ray.rs:42-44"""
    x, y, out = _a(in_vector, 4), _a(normal_vector, 4), np.zeros(4, dtype=f32)
    L.lib().rtc_reflect(_p(x), _p(y), _p(out))
    return out


# ------------------------------------------------- matrix.rs, transformations.rs
def identity_4x4():
    return np.eye(4, dtype=f32)


def _mat(fn, *args):
    out = np.zeros(16, dtype=f32)
    fn(*args, _p(out))
    return out.reshape(4, 4)


def translation(x, y, z):
    return _mat(L.lib().rtc_translation, float(f32(x)), float(f32(y)), float(f32(z)))


def scaling(x, y, z):
    return _mat(L.lib().rtc_scaling, float(f32(x)), float(f32(y)), float(f32(z)))


def rotation_x(radians):
    return _mat(L.lib().rtc_rotation_x, float(f32(radians)))


def rotation_y(radians):
    return _mat(L.lib().rtc_rotation_y, float(f32(radians)))


def rotation_z(radians):
    return _mat(L.lib().rtc_rotation_z, float(f32(radians)))


def shearing(x_y, x_z, y_x, y_z, z_x, z_y):
    return _mat(L.lib().rtc_shearing, *[float(f32(v)) for v in (x_y, x_z, y_x, y_z, z_x, z_y)])


def view_transform(frm, to, approximate_up):
    a, b, c = _a(frm, 4), _a(to, 4), _a(approximate_up, 4)
    return _mat(L.lib().rtc_view_transform, _p(a), _p(b), _p(c))


def mat_mul(a, b):
    x, y = _a(a, 16), _a(b, 16)
    return _mat(L.lib().rtc_mat_mul, _p(x), _p(y))


def chain(*ms):
    """`a * b * c` as Rust evaluates it: ((a*b)*c)."""
    out = ms[0]
    for m in ms[1:]:
        out = mat_mul(out, m)
    return out


def mat_vec(a, v):
    x, y, out = _a(a, 16), _a(v, 4), np.zeros(4, dtype=f32)
    L.lib().rtc_mat_vec(_p(x), _p(y), _p(out))
    return out


def _sq(a):
    a = np.asarray(a, dtype=f32)
    return a.shape[0], _a(a)


def transpose(a):
    n, x = _sq(a)
    out = np.zeros(n * n, dtype=f32)
    L.lib().rtc_mat_transpose(_p(x), n, _p(out))
    return out.reshape(n, n)


def determinant(a):
    n, x = _sq(a)
    return f32(L.lib().rtc_mat_determinant(_p(x), n))


def submatrix(a, row, col):
    n, x = _sq(a)
    out = np.zeros((n - 1) * (n - 1), dtype=f32)
    L.lib().rtc_mat_submatrix(_p(x), n, row, col, _p(out))
    return out.reshape(n - 1, n - 1)


def minor(a, row, col):
    n, x = _sq(a)
    return f32(L.lib().rtc_mat_minor(_p(x), n, row, col))


def cofactor(a, row, col):
    n, x = _sq(a)
    return f32(L.lib().rtc_mat_cofactor(_p(x), n, row, col))


def inverse(a):
    n, x = _sq(a)
    out = np.zeros(n * n, dtype=f32)
    L.check(L.lib().rtc_mat_inverse(_p(x), n, _p(out)))
    return out.reshape(n, n)


# ------------------------------------------------------------------- pattern/*.rs
class Pattern:
    """A procedural pattern: two colours and a pattern->object transform (pattern/pattern.rs:8-26).
    color_at_world / color_at_object evaluate on the device."""

    def __init__(self, kind, a=(1.0, 1.0, 1.0), b=(0.0, 0.0, 0.0), transform=None, uv_mapping=0, uv=()):
        self.kind = kind
        self.a, self.b = tuple(float(c) for c in a), tuple(float(c) for c in b)
        self.transform = identity_4x4() if transform is None else np.asarray(transform, dtype=f32).reshape(4, 4)
        self.uv_mapping, self.uv = uv_mapping, list(uv)  # TextureMap / CubicMap

    def set_transformation(self, t):
        """pattern.rs:20-22; the inverse is taken when the pattern is flattened."""
        self.transform = np.asarray(t, dtype=f32).reshape(4, 4)

    def transformation_inverse(self):
        return np.array(list(self._c().inv), dtype=f32).reshape(4, 4)

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
        p = L.rtc_pattern()
        a, b, t = _a(self.a, 3), _a(self.b, 3), _a(self.transform, 16)
        if self.uv:
            arr = self._uv_array(L.rtc_uv_pattern)
            L.check(L.lib().rtc_texture_map_init(C.byref(p), int(self.uv_mapping), arr, len(self.uv), _p(t)))
            return p
        L.check(L.lib().rtc_pattern_init(C.byref(p), self.kind, _p(a), _p(b), _p(t)))
        return p

    def color_at_object(self, world_points, shape=None, device=0):
        """Pattern::color_at_object (pattern.rs:15-19), batched: (n, 4) world points -> (n, 3)."""
        pts = np.ascontiguousarray(np.asarray(world_points, dtype=f32).reshape(-1, 4))
        out = np.zeros((pts.shape[0], 3), dtype=f32)
        p = self._c()
        o = shape._c() if shape is not None else None
        L.check(L.lib().rtc_pattern_color_at(C.byref(p), C.byref(o) if o is not None else None, _p(pts), pts.shape[0],
                                             device, _p(out)))
        return out

    def color_at_world(self, points, device=0):
        """Pattern::color_at_world: the pattern-space lookup itself (identity object and pattern transforms)."""
        return Pattern(self.kind, self.a, self.b, None, self.uv_mapping, self.uv).color_at_object(points, None, device)


def Stripes(a=(1.0, 1.0, 1.0), b=(0.0, 0.0, 0.0), transform=None):
    """Stripes::new(a, b) -- pattern/stripes.rs:17-24 (default: white, black)"""
    return Pattern(L.RTC_PATTERN_STRIPES, a, b, transform)


def Gradient(a=(1.0, 1.0, 1.0), b=(0.0, 0.0, 0.0), transform=None):
    """Gradient::new -- pattern/gradient.rs:16-23"""
    return Pattern(L.RTC_PATTERN_GRADIENT, a, b, transform)


def Rings(a=(1.0, 1.0, 1.0), b=(0.0, 0.0, 0.0), transform=None):
    """Rings::new -- pattern/rings.rs:16-22"""
    return Pattern(L.RTC_PATTERN_RINGS, a, b, transform)


def Checkers(a=(1.0, 1.0, 1.0), b=(0.0, 0.0, 0.0), transform=None):
    """Checkers::new -- pattern/checkers.rs:16-22"""
    return Pattern(L.RTC_PATTERN_CHECKERS, a, b, transform)


def Sine2D(a=(1.0, 1.0, 1.0), b=(0.0, 0.0, 0.0), transform=None):
    """Sine2D::new -- pattern/sine_2d.rs:16-23"""
    return Pattern(L.RTC_PATTERN_SINE2D, a, b, transform)


# ------------------------------------------------------------------- pattern/uv.rs
class UVCheckers:
    """UVCheckers::new(width, height, a, b) -- pattern/uv.rs:28-36 (default 1, 1, white, black)"""

    def __init__(self, width=1.0, height=1.0, a=(1.0, 1.0, 1.0), b=(0.0, 0.0, 0.0)):
        self.width, self.height, self.a, self.b = width, height, tuple(a), tuple(b)

    def _c(self):
        u = L.rtc_uv_pattern()
        u.kind, u.width, u.height = L.RTC_UV_CHECKERS, float(f32(self.width)), float(f32(self.height))
        u.colors[0][:] = [float(f32(c)) for c in self.a]
        u.colors[1][:] = [float(f32(c)) for c in self.b]
        return u


class AlignCheck:
    """AlignCheck::new(main, ul, ur, bl, br) -- pattern/uv.rs:135-143 (default white, red, yellow, green, cyan)"""

    def __init__(self, main=(1, 1, 1), ul=(1, 0, 0), ur=(1, 1, 0), bl=(0, 1, 0), br=(0, 1, 1)):
        self.colors = [tuple(c) for c in (main, ul, ur, bl, br)]

    def _c(self):
        u = L.rtc_uv_pattern()
        u.kind = L.RTC_UV_ALIGN_CHECK
        for k, col in enumerate(self.colors):
            u.colors[k][:] = [float(f32(c)) for c in col]
        return u


class UVImage:
    """UVImage::new(canvas) -- pattern/uv.rs:351-355; canvas: a Canvas or an (h, w, 3) f32 array"""

    def __init__(self, canvas):
        data = canvas.data if hasattr(canvas, "data") and not isinstance(canvas, np.ndarray) else canvas
        self.canvas = np.ascontiguousarray(data, dtype=f32)

    def _c(self):
        u = L.rtc_uv_pattern()
        u.kind = L.RTC_UV_IMAGE
        u.image_height, u.image_width = self.canvas.shape[0], self.canvas.shape[1]
        u.image_rgb = _p(self.canvas)
        return u


class SphericalMap:  # pattern/uv.rs:91-105
    kind = L.RTC_MAP_SPHERICAL


class PlanarMap:  # pattern/uv.rs:180-186
    kind = L.RTC_MAP_PLANAR


class CylindricalMap:  # pattern/uv.rs:188-198
    kind = L.RTC_MAP_CYLINDRICAL


def TextureMap(uv_pattern, uv_mapping, transform=None):
    """TextureMap::new(uv_pattern, uv_mapping) -- pattern/uv.rs:68-76"""
    return Pattern(L.RTC_PATTERN_TEXTURE_MAP, transform=transform, uv_mapping=uv_mapping.kind, uv=[uv_pattern])


def CubicMap(front, back, left, right, up, down, transform=None):
    """CubicMap::new(front, back, left, right, up, down) -- pattern/uv.rs:207-231"""
    return Pattern(L.RTC_PATTERN_CUBE_MAP, transform=transform, uv=[front, back, left, right, up, down])


def canvas_from_ppm(text):
    """canvas_from_ppm (canvas.rs:120-197): P3 text -> Canvas"""
    if isinstance(text, str):
        text = text.encode()
    w, h, rgb = C.c_uint32(), C.c_uint32(), C.c_void_p()
    L.check(L.lib().rtc_canvas_from_ppm(text, len(text), C.byref(w), C.byref(h), C.byref(rgb)))
    try:
        n = w.value * h.value * 3
        data = np.ctypeslib.as_array(C.cast(rgb, L.FP), shape=(max(n, 1),))[:n].copy()
    finally:
        L.lib().rtc_free(rgb)
    return Canvas(w.value, h.value, data.reshape(h.value, w.value, 3))


# ------------------------------------------------------------------ material.rs
class Material:
    """Material::builder() with the reference's defaults (material.rs:18-51)."""

    FIELDS = ("ambient", "diffuse", "specular", "shininess", "reflective", "transparency", "refractive_index")

    def __init__(self, color=(1.0, 1.0, 1.0), ambient=0.1, diffuse=0.9, specular=0.9, shininess=200.0,
                 reflective=0.0, transparency=0.0, refractive_index=1.0, pattern=None):
        self.color = tuple(float(c) for c in color)
        self.ambient, self.diffuse, self.specular, self.shininess = ambient, diffuse, specular, shininess
        self.reflective, self.transparency, self.refractive_index = reflective, transparency, refractive_index
        self.pattern = pattern

    def copy(self, **changes):
        m = Material(self.color, *[getattr(self, k) for k in self.FIELDS], pattern=self.pattern)
        for k, v in changes.items():
            setattr(m, k, v)
        return m

    def _c(self):
        m = L.rtc_material()
        L.lib().rtc_material_default(C.byref(m))
        m.color[:] = [float(f32(c)) for c in self.color]
        for k in self.FIELDS:
            setattr(m, k, float(f32(getattr(self, k))))
        if self.pattern is not None:
            m.pattern = self.pattern._c()
        return m


def glass():
    """constants.rs:12-17"""
    return Material(transparency=1.0, refractive_index=1.52)


def metal():
    """constants.rs:50-62"""
    return Material(color=(0.5, 0.5, 0.5), ambient=1.0, diffuse=0.6, reflective=0.1, specular=0.4, shininess=10.0)


# ------------------------------------------------------------------- shape/*.rs
class Shape:
    """BaseShape + concrete kind (shape/base_shape.rs:13-20)."""

    def __init__(self, kind, transform=None, material=None, casts_shadow=True, minimum_y=-np.inf,
                 maximum_y=np.inf, closed=False):
        self.kind = kind
        self.transform = identity_4x4() if transform is None else np.asarray(transform, dtype=f32).reshape(4, 4)
        self.material = Material() if material is None else material
        self.casts_shadow = casts_shadow
        self.minimum_y, self.maximum_y, self.closed = minimum_y, maximum_y, closed
        self.points = None  # Triangle: (p1, p2, p3)

    # reference setter names
    def set_transformation(self, t):
        self.transform = np.asarray(t, dtype=f32).reshape(4, 4)

    def set_material(self, m):
        self.material = m

    def set_casts_shadow(self, flag):
        self.casts_shadow = flag

    def transformation_inverse(self):
        return np.array(list(self._c().inv), dtype=f32).reshape(4, 4)

    def _c(self):
        o = L.rtc_object()
        t = _a(self.transform, 16)
        m = self.material._c()
        L.check(L.lib().rtc_object_init(C.byref(o), self.kind, _p(t), C.byref(m)))
        o.casts_shadow = int(bool(self.casts_shadow))
        o.closed = int(bool(self.closed))
        o.min_y = float(f32(self.minimum_y))
        o.max_y = float(f32(self.maximum_y))
        if self.points is not None:
            for name, p in zip(("p1", "p2", "p3"), self.points):
                getattr(o, name)[:] = [float(v) for v in p[:3]]
        return o

    def local_intersect(self, origins, directions, device=0):
        """Shape::local_intersect, batched on the device: (n, 4) object-space rays -> list of per-ray distance
        lists in the reference's push order."""
        o = np.ascontiguousarray(np.asarray(origins, dtype=f32).reshape(-1, 4))
        d = np.ascontiguousarray(np.asarray(directions, dtype=f32).reshape(-1, 4))
        if o.shape != d.shape:
            raise ValueError("origins and directions differ in shape: %r, %r" % (o.shape, d.shape))
        ts = np.zeros((o.shape[0], 4), dtype=f32)
        counts = np.zeros(o.shape[0], dtype=np.int32)
        c = self._c()
        L.check(L.lib().rtc_local_intersect(C.byref(c), _p(o), _p(d), o.shape[0], device, _p(ts),
                                            counts.ctypes.data_as(C.POINTER(C.c_int32))))
        return [list(ts[i, :counts[i]]) for i in range(o.shape[0])]

    def normal_at(self, world_points, device=0):
        """Shape::normal_at (shape.rs:72-154), batched on the device: (n, 4) points -> (n, 4) vectors."""
        p = np.ascontiguousarray(np.asarray(world_points, dtype=f32).reshape(-1, 4))
        out = np.zeros_like(p)
        c = self._c()
        L.check(L.lib().rtc_normal_at(C.byref(c), _p(p), p.shape[0], device, _p(out)))
        return out


def Sphere(transform=None, material=None, **kw):
    """Sphere::build(transform, material) -- shape/sphere.rs:23-28"""
    return Shape(L.RTC_SPHERE, transform, material, **kw)


def Plane(transform=None, material=None, **kw):
    """Plane::build -- shape/plane.rs:21-26"""
    return Shape(L.RTC_PLANE, transform, material, **kw)


def Cube(transform=None, material=None, **kw):
    """Cube::build -- shape/cube.rs:31-36"""
    return Shape(L.RTC_CUBE, transform, material, **kw)


def Cylinder(transform=None, material=None, **kw):
    """Cylinder::build; minimum_y / maximum_y / closed are its pub fields (shape/cylinder.rs:14-19)"""
    return Shape(L.RTC_CYLINDER, transform, material, **kw)


def Triangle(p1, p2, p3, transform=None, material=None, **kw):
    """Triangle::new(p1, p2, p3) -- shape/triangle.rs:19-33.  e1 / e2 / normal: see triangle_fields()."""
    t = Shape(L.RTC_TRIANGLE, transform, material, **kw)
    t.points = [_a(p, 4).copy() for p in (p1, p2, p3)]
    return t


def triangle_fields(tri):
    """(e1, e2, normal) exactly as Triangle::new derives them (triangle.rs:20-22)."""
    p1, p2, p3 = [_a(p[:3], 3) for p in tri.points]
    out = [np.zeros(3, dtype=f32) for _ in range(3)]
    L.lib().rtc_triangle_fields(_p(p1), _p(p2), _p(p3), *[_p(x) for x in out])
    return [np.append(x, f32(0.0)) for x in out]


def SmoothTriangle(p1, p2, p3, n1, n2, n3, transform=None, material=None, **kw):
    """SmoothTriangle::new -- shape/smooth_triangle.rs:17-26.  Its local_intersect returns intersections whose
    object is the inner flat Triangle (:37-39), so Camera::render shades it as one; that is what this does too.
    The vertex normals are kept for local_norm_at_uv(), which restates :41-43 for hand-made hits."""
    t = Triangle(p1, p2, p3, transform, material, **kw)
    t.normals = [_a(n, 4).copy() for n in (n1, n2, n3)]

    def local_norm_at_uv(u, v):
        u, v = f32(u), f32(v)
        n1_, n2_, n3_ = t.normals
        return n2_ * u + n3_ * v + n1_ * (f32(1.0) - u - v)
    t.local_norm_at_uv = local_norm_at_uv
    return t


# ------------------------------------------------- bounding_box.rs / shape/group.rs
class BoundingBox:
    """bounding_box.rs:7-11 over the library's host helpers (rtc_bounds_*)."""

    def __init__(self, mn, mx):
        self.min, self.max = _a(mn, 4).copy(), _a(mx, 4).copy()

    @staticmethod
    def empty():
        mn, mx = np.zeros(4, dtype=f32), np.zeros(4, dtype=f32)
        L.lib().rtc_bounds_empty(_p(mn), _p(mx))
        return BoundingBox(mn, mx)

    @staticmethod
    def with_bounds(mn, mx):
        return BoundingBox(mn, mx)

    def add_bounding_box(self, other):
        L.lib().rtc_bounds_add(_p(self.min), _p(self.max), _p(other.min), _p(other.max))

    def contains_bounding_box(self, other):
        return bool(L.lib().rtc_bounds_contains(_p(self.min), _p(self.max), _p(other.min), _p(other.max)))

    def transform(self, m):
        mm, mn, mx = _a(m, 16), np.zeros(4, dtype=f32), np.zeros(4, dtype=f32)
        L.lib().rtc_bounds_transform(_p(self.min), _p(self.max), _p(mm), _p(mn), _p(mx))
        return BoundingBox(mn, mx)

    def split(self):
        out = [np.zeros(4, dtype=f32) for _ in range(4)]
        L.lib().rtc_bounds_split(_p(self.min), _p(self.max), *[_p(x) for x in out])
        return BoundingBox(out[0], out[1]), BoundingBox(out[2], out[3])


def _shape_bounds(shape, parent_space):
    mn, mx = np.zeros(4, dtype=f32), np.zeros(4, dtype=f32)
    t = _a(shape.transform, 16) if parent_space else None
    if shape.points is not None:
        p1, p2, p3 = [_a(p[:3], 3) for p in shape.points]
        L.lib().rtc_triangle_bounds(_p(p1), _p(p2), _p(p3), _p(t) if t is not None else None, _p(mn), _p(mx))
        return BoundingBox(mn, mx)
    L.check(L.lib().rtc_shape_bounds(shape.kind, float(f32(shape.minimum_y)), float(f32(shape.maximum_y)),
                                     _p(t) if t is not None else None, _p(mn), _p(mx)))
    return BoundingBox(mn, mx)


# Shape::bounding_box / parent_space_bounding_box (shape.rs:23,162-164); divide is a no-op for leaves (:167)
Shape.bounding_box = lambda self: _shape_bounds(self, False)


def _parent_space_bounds_cached(self):
    """divide() asks every leaf for its parent-space box once per level of the tree; the answer is a pure function
    of what is keyed on here, so it is computed once (a 100 k-triangle mesh otherwise spends its build time here)."""
    pts = self.points
    key = (self.transform.tobytes(), self.kind, float(self.minimum_y), float(self.maximum_y),
           None if pts is None else tuple(np.asarray(q, dtype=f32).tobytes() for q in pts))
    hit = getattr(self, "_psbb", None)
    if hit is None or hit[0] != key:
        hit = (key, _shape_bounds(self, True))
        self._psbb = hit
    return BoundingBox(hit[1].min.copy(), hit[1].max.copy())


Shape.parent_space_bounding_box = _parent_space_bounds_cached
Shape.divide = lambda self, threshold: None
Shape.transformation = lambda self: self.transform


class GroupShape:
    """shape/group.rs.  add_child / set_transformation bake the group's transform into the children, so a world
    with groups flattens to its leaf shapes plus one bounding box per group (what the device path consumes)."""

    def __init__(self):
        self.transform = identity_4x4()
        self._t_inverse = identity_4x4()  # BaseShape::default(): not an inversion
        self.children = []
        self._cached_box = None           # cached_bounding_box (:15): filled on first use, never invalidated
        self.casts_shadow = True          # the group's own BaseShape flag (base_shape.rs:31): settable, never consulted

    @staticmethod
    def with_children(children):
        """group.rs:23-27: adopts the children as they are (nothing is re-baked)."""
        g = GroupShape()
        g.children = list(children)
        return g

    def get_children(self):
        return self.children

    def transformation(self):
        return self.transform

    def add_child(self, child):
        """group.rs:39-44"""
        old_child_transform = child.transformation().copy()
        child.set_transformation(mat_mul(self.transform, old_child_transform))
        self.children.append(child)

    def set_transformation(self, t):
        """group.rs:101-114"""
        t = np.asarray(t, dtype=f32).reshape(4, 4)
        if self.children:
            child_transformer = mat_mul(t, self._t_inverse)
            for c in self.children:
                old_child_transform = c.transformation().copy()
                c.set_transformation(mat_mul(child_transformer, old_child_transform))
        self.transform = t.copy()
        self._t_inverse = inverse(t)

    def set_material(self, m):
        """group.rs:96-100"""
        for c in self.children:
            c.set_material(m.copy())

    def casts_shadow_flag(self):
        return self.casts_shadow

    def set_casts_shadow(self, flag):
        """shape.rs:45-47 on a group: sets the flag of the GROUP's own BaseShape, which nothing ever reads -- intersections
        carry the leaf that was hit (group.rs:119-133), and World::is_shadowed asks that leaf (world.rs:117).  The
        children keep their own flags; a no-op for rendering, as in the reference."""
        self.casts_shadow = bool(flag)

    def bounding_box(self):
        """group.rs:138-151"""
        if self._cached_box is None:
            b = BoundingBox.empty()
            for child in self.children:
                b.add_bounding_box(child.parent_space_bounding_box())
            self._cached_box = b
        return self._cached_box

    def parent_space_bounding_box(self):
        return self.bounding_box()  # group.rs:153-155

    def _partition_children(self):
        """group.rs:46-64"""
        left_bounds, right_bounds = self.bounding_box().split()
        left, right, keep = [], [], []
        for c in self.children:
            child_bounds = c.parent_space_bounding_box()
            if left_bounds.contains_bounding_box(child_bounds):
                left.append(c)
            elif right_bounds.contains_bounding_box(child_bounds):
                right.append(c)
            else:
                keep.append(c)
        self.children = keep
        return left, right

    def _make_subgroup(self, new_group_children):
        """group.rs:66-73"""
        if len(new_group_children) == 1:
            self.children.append(new_group_children[0])
        else:
            self.children.append(GroupShape.with_children(new_group_children))

    def divide(self, threshold):
        """group.rs:157-172"""
        if threshold <= len(self.children):
            left, right = self._partition_children()
            if left:
                self._make_subgroup(left)
            if right:
                self._make_subgroup(right)
        for child in self.children:
            child.divide(threshold)

    def leaves(self):
        out = []
        for c in self.children:
            out.extend(c.leaves() if isinstance(c, GroupShape) else [c])
        return out


def Cone(transform=None, material=None, **kw):
    """Cone::build; minimum_y / maximum_y / closed are its pub fields (shape/cone.rs:12-17)"""
    return Shape(L.RTC_CONE, transform, material, **kw)


# ------------------------------------------------------------------- light/*.rs
class PointLight:
    """PointLight::new(position, intensity) -- light/point_light.rs:12-19"""

    def __init__(self, position, intensity):
        self.position, self.intensity = _a(position, 4), _a(intensity, 3)

    def _c(self):
        l = L.rtc_light()
        L.lib().rtc_point_light(_p(self.position), _p(self.intensity), C.byref(l))
        return l


class RectangleLight:
    """RectangleLight::new(intensity, corner, u_vec, u_steps, v_vec, v_steps, jitter_fn_opt)
    -- light/rectangle_light.rs:33-58.

    jitter: ("constant", c) mirrors test/utils.rs constant_jitter();
            ("hashed", seed) stands in for jitter_fn_opt = None (thread_rng);
            ("cycle", [values]) mirrors test/utils.rs hardcoded_jitter(): state carried from call to call, so only
            World.intensity_at and point_on_light -- each answered as by a freshly built light -- accept it.
            A Python callable (the closure form) cannot run on the device -> RtcError(UNSUPPORTED).
    """

    def __init__(self, intensity, corner, u_vec, u_steps, v_vec, v_steps, jitter=("hashed", 0x5EED5EED)):
        self.intensity, self.corner = _a(intensity, 3), _a(corner, 4)
        self.u_vec, self.v_vec = _a(u_vec, 4), _a(v_vec, 4)
        self.u_steps, self.v_steps, self.jitter = int(u_steps), int(v_steps), jitter

    def _c(self):
        l = L.rtc_light()
        seq = None
        if callable(self.jitter):
            mode, const, seed = 1, 0.0, 0  # closure: rejected by the library
        else:
            kind, arg = self.jitter
            if kind == "constant":
                mode, const, seed = L.RTC_JITTER_CONSTANT, float(f32(arg)), 0
            elif kind == "hashed":
                mode, const, seed = L.RTC_JITTER_HASHED, 0.0, int(arg) & 0xFFFFFFFF
            elif kind == "cycle":
                mode, const, seed, seq = L.RTC_JITTER_SEQUENCE, 0.0, 0, _a(list(arg))
            else:
                mode, const, seed = 1, 0.0, 0
        L.check(L.lib().rtc_rectangle_light(_p(self.intensity), _p(self.corner), _p(self.u_vec), self.u_steps,
                                            _p(self.v_vec), self.v_steps, mode, const, seed, C.byref(l)))
        if seq is not None:
            L.check(L.lib().rtc_light_set_jitter_sequence(C.byref(l), _p(seq), len(seq)))
        return l

    def point_on_light(self, cells_uv, device=0):
        """RectangleLight::point_on_light (rectangle_light.rs:60-66) on the device for (n, 2) cell pairs, each as the first
        call on a freshly built light -> (n, 4) points."""
        uv = np.ascontiguousarray(np.asarray(cells_uv, dtype=np.int32).reshape(-1, 2))
        out = np.zeros((uv.shape[0], 4), dtype=f32)
        l = self._c()
        L.check(L.lib().rtc_point_on_light(C.byref(l), uv.ctypes.data_as(C.POINTER(C.c_int32)), uv.shape[0], device, _p(out)))
        return out

    # fields the reference exposes after construction
    @property
    def position(self):
        return np.array(list(self._c().position), dtype=f32)

    @property
    def cell_u_vec(self):
        return np.array(list(self._c().u_vec), dtype=f32)

    @property
    def cell_v_vec(self):
        return np.array(list(self._c().v_vec), dtype=f32)

    @property
    def cells(self):
        return self.u_steps * self.v_steps


# --------------------------------------------------------------------- world.rs
class _CScene:
    """Keeps the ctypes arrays alive for the lifetime of a call.  World.objects may hold GroupShapes: the tree is
    written out as its leaves in depth-first order plus one rtc_group (leaf run + bounding box) per group."""

    def __init__(self, world):
        leaves, groups = [], []

        def walk(node):
            if isinstance(node, GroupShape):
                rec = L.rtc_group()
                rec.first_object = len(leaves)
                box = node.bounding_box()
                rec.bounds_min[:] = [float(v) for v in box.min[:3]]
                rec.bounds_max[:] = [float(v) for v in box.max[:3]]
                groups.append(rec)
                for child in node.children:
                    walk(child)
                rec.n_objects = len(leaves) - rec.first_object
            else:
                leaves.append(node)
        for o in world.objects:
            walk(o)
        n = len(leaves)
        self.leaves = leaves
        self.objects = (L.rtc_object * max(n, 1))()
        for i, o in enumerate(leaves):
            self.objects[i] = o._c()
        self.groups = (L.rtc_group * max(len(groups), 1))(*groups)
        self.light = world.light._c() if world.light is not None else None
        self.scene = L.rtc_scene()
        self.scene.n_objects = n
        self.scene.objects = C.cast(self.objects, C.POINTER(L.rtc_object))
        self.scene.light = C.pointer(self.light) if self.light is not None else None
        self.scene.n_groups = len(groups)
        self.scene.groups = C.cast(self.groups, C.POINTER(L.rtc_group))


class World:
    """World { objects, light } -- world.rs:18-21"""

    def __init__(self, objects=(), light=None):
        self.objects = list(objects)
        self.light = light

    def _c(self):
        return _CScene(self)

    def validate(self, camera=None):
        """rtc_scene_validate: the checks rtc_ctx_set_scene would make, without a GPU (raises RtcError)."""
        cs = self._c()
        L.check(L.lib().rtc_scene_validate(C.byref(cs.scene), C.byref(camera._cam) if camera is not None else None))

    def color_at(self, origins, directions, depth, device=0):
        """World::color_at for a batch of rays (world.rs:88-101); (n,4),(n,4) -> (n,3)."""
        o = np.ascontiguousarray(np.asarray(origins, dtype=f32).reshape(-1, 4))
        d = np.ascontiguousarray(np.asarray(directions, dtype=f32).reshape(-1, 4))
        out = np.zeros((o.shape[0], 3), dtype=f32)
        cs = self._c()
        L.check(L.lib().rtc_color_at(C.byref(cs.scene), _p(o), _p(d), o.shape[0], int(depth), device, _p(out)))
        return out

    def intensity_at(self, points, device=0):
        """Light::intensity_at for a batch of points (light/light.rs:10)."""
        p = np.ascontiguousarray(np.asarray(points, dtype=f32).reshape(-1, 4))
        out = np.zeros(p.shape[0], dtype=f32)
        cs = self._c()
        L.check(L.lib().rtc_intensity_at(C.byref(cs.scene), _p(p), p.shape[0], device, _p(out)))
        return out

    def is_shadowed(self, light_positions, points, device=0):
        """World::is_shadowed for a batch (world.rs:104-119)."""
        l = np.ascontiguousarray(np.asarray(light_positions, dtype=f32).reshape(-1, 4))
        p = np.ascontiguousarray(np.asarray(points, dtype=f32).reshape(-1, 4))
        out = np.zeros(p.shape[0], dtype=np.int32)
        cs = self._c()
        L.check(L.lib().rtc_is_shadowed(C.byref(cs.scene), _p(l), _p(p), p.shape[0], device,
                                        out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out.astype(bool)


def default_world():
    """World::default() -- world.rs:32-48"""
    s1 = Sphere(identity_4x4(), Material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2))
    s2 = Sphere(scaling(0.5, 0.5, 0.5), Material())
    return World([s1, s2], PointLight(point(-10.0, 10.0, -10.0), color(1, 1, 1)))


# -------------------------------------------------------------------- canvas.rs
class Canvas:
    """Canvas (canvas.rs:6-10) over an (h, w, 3) float32 array."""

    def __init__(self, width, height, data=None):
        self.width, self.height = int(width), int(height)
        self.data = np.zeros((self.height, self.width, 3), dtype=f32) if data is None else data

    def write_pixel(self, x, y, c):
        if x <= self.width and y <= self.height:  # canvas.rs:27 (sic)
            self.data[y, x] = c

    def pixel_at(self, x, y):
        return self.data[y, x]

    def to_ppm(self):
        """Canvas::to_ppm (canvas.rs:58-96) -> bytes."""
        img = np.ascontiguousarray(self.data, dtype=f32)
        text, n = C.c_void_p(), C.c_uint64()
        L.check(L.lib().rtc_to_ppm(_p(img), self.width, self.height, C.byref(text), C.byref(n)))
        try:
            return C.string_at(text, n.value)
        finally:
            L.lib().rtc_free(text)


# -------------------------------------------------------------------- camera.rs
class Camera:
    """Camera::new(width_pixels, height_pixels, field_of_view, transform) -- camera.rs:23-56"""

    def __init__(self, width_pixels, height_pixels, field_of_view, transform):
        t = _a(transform, 16)
        self._cam = L.rtc_camera()
        L.check(L.lib().rtc_camera_new(int(width_pixels), int(height_pixels), float(f32(field_of_view)), _p(t),
                                       C.byref(self._cam)))
        self.width, self.height = int(width_pixels), int(height_pixels)
        self.field_of_view = f32(field_of_view)
        self.transform = np.asarray(transform, dtype=f32).reshape(4, 4).copy()
        self.last_stats = None

    @property
    def pixel_size(self):
        return f32(self._cam.pixel_size)

    @property
    def half_width(self):
        return f32(self._cam.half_width)

    @property
    def half_height(self):
        return f32(self._cam.half_height)

    @property
    def transform_inverse(self):
        return np.array(list(self._cam.inv), dtype=f32).reshape(4, 4)

    def ray_for_pixel(self, x, y):
        o, d = np.zeros(4, dtype=f32), np.zeros(4, dtype=f32)
        L.lib().rtc_ray_for_pixel(C.byref(self._cam), int(x), int(y), _p(o), _p(d))
        return o, d

    def render(self, world, reflection_recursion_depth, device=0, devices=None, quantize=False, band_rows=0, out=None):
        """Camera::render (camera.rs:76-91) on the MI355X -> Canvas.

        devices: the GPUs to split the image over (rtc_render_ex; default: [device]).  quantize: return the (h, w, 3) u8
        array of scale_color'd channels (canvas.rs:39-43) instead of a Canvas.  out: a caller-provided array to fill."""
        devs = [int(device)] if devices is None else [int(d) for d in devices]
        dtype = np.uint8 if quantize else f32
        img = np.zeros((self.height, self.width, 3), dtype=dtype) if out is None else out
        if not isinstance(img, np.ndarray) or img.dtype != dtype or img.shape != (self.height, self.width, 3) or not img.flags["C_CONTIGUOUS"]:
            # (an explicit check, not an assert: the raw pointer goes to native code, which writes height * width * 3 values)
            raise ValueError("out must be a C-contiguous (%d, %d, 3) array of %s" % (self.height, self.width, np.dtype(dtype).name))
        stats = L.rtc_stats()
        cs = world._c()
        _seam_follows_environment()
        arr = (C.c_int32 * len(devs))(*devs)
        opts = L.rtc_opts(arr, len(devs), int(band_rows), 1 if quantize else 0, 0)
        L.check(L.lib().rtc_render_ex(C.byref(cs.scene), C.byref(self._cam), int(reflection_recursion_depth), C.byref(opts),
                                      img.ctypes.data_as(C.c_void_p), C.byref(stats)))
        self.last_stats = {"rays": int(stats.rays), "shaded_hits": int(stats.shaded_hits),
                           "pixels": int(stats.pixels), "kernel_ms": float(stats.kernel_ms),
                           "culled_shadow_rays": int(stats.culled_shadow_rays), "launches": int(stats.launches),
                           "call_ms": float(stats.gather_ms), "flags": int(stats.flags)}
        return img if quantize else Canvas(self.width, self.height, img)


_seam_env = {}


def _seam_follows_environment():
    """The library reads its RTC_AMD_* switches once, when a context is created, and rtc_render_ex keeps its contexts
    between calls (rtc_render_release drops them).  Scripts and tests flip a switch between two renders of one process and
    expect the second to see it: this mirror drops the seam's contexts whenever the process's RTC_AMD_* environment
    differs from what it was at its previous render.  (A C caller does the same with rtc_render_release.)"""
    import os
    lib = L.lib()
    env = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("RTC_AMD_")))
    if _seam_env.get(id(lib), env) != env:
        lib.rtc_render_release()
    _seam_env[id(lib)] = env


def powf(x, y, device=0):
    """f32::powf on the device (phong_lighting.rs:56)."""
    a, b = _a(x), _a(y)
    out = np.zeros(a.size, dtype=f32)
    L.check(L.lib().rtc_powf(_p(a), _p(b), a.size, device, _p(out)))
    return out


def cosf(x, device=0):
    """f32::cos on the device (pattern/sine_2d.rs:40)."""
    a = _a(x)
    out = np.zeros(a.size, dtype=f32)
    L.check(L.lib().rtc_cosf(_p(a), a.size, device, _p(out)))
    return out


def atan2f(y, x, device=0):
    """f32::atan2 on the device (pattern/uv.rs:108)."""
    a, b = _a(y), _a(x)
    out = np.zeros(a.size, dtype=f32)
    L.check(L.lib().rtc_atan2f(_p(a), _p(b), a.size, device, _p(out)))
    return out


def acosf(x, device=0):
    """f32::acos on the device (pattern/uv.rs:101)."""
    a = _a(x)
    out = np.zeros(a.size, dtype=f32)
    L.check(L.lib().rtc_acosf(_p(a), a.size, device, _p(out)))
    return out


def atan2f_host(y, x):
    a, b = _a(y), _a(x)
    out = np.zeros(a.size, dtype=f32)
    L.lib().rtc_atan2f_host(_p(a), _p(b), a.size, _p(out))
    return out


def acosf_host(x):
    a = _a(x)
    out = np.zeros(a.size, dtype=f32)
    L.lib().rtc_acosf_host(_p(a), a.size, _p(out))
    return out


def cosf_host(x):
    """Host compile of the device cosf restatement (diagnostic only)."""
    a = _a(x)
    out = np.zeros(a.size, dtype=f32)
    L.lib().rtc_cosf_host(_p(a), a.size, _p(out))
    return out


def powf_host(x, y):
    """Host compile of the device powf restatement (diagnostic only)."""
    a, b = _a(x), _a(y)
    out = np.zeros(a.size, dtype=f32)
    L.lib().rtc_powf_host(_p(a), _p(b), a.size, _p(out))
    return out


def device_count():
    return int(L.lib().rtc_device_count())
