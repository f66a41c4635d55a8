"""ctypes declarations for librtc_amd.so -- a 1:1 transcription of include/rtc.h.

The library must have been built (python -m ray_tracer_challenge_amd.build, or
__graft_entry__.build()).  There is no Python or CPU fallback: if the shared
object is missing the import fails, and every device entry point returns
RTC_ERR_NO_DEVICE when no GPU is visible.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RTC_AMD_LIB: development override used by tools/ab.py to load a kernel variant
LIB_PATH = os.environ.get("RTC_AMD_LIB") or os.path.join(HERE, "librtc_amd.so")

RTC_OK = 0
RTC_ERR_INVALID_ARG, RTC_ERR_UNSUPPORTED, RTC_ERR_NO_LIGHT, RTC_ERR_DEVICE, RTC_ERR_NO_DEVICE = -1, -2, -3, -4, -5
RTC_SPHERE, RTC_PLANE, RTC_CUBE, RTC_CYLINDER, RTC_CONE, RTC_TRIANGLE = 0, 1, 2, 3, 4, 5
(RTC_PATTERN_NONE, RTC_PATTERN_STRIPES, RTC_PATTERN_GRADIENT, RTC_PATTERN_RINGS, RTC_PATTERN_CHECKERS,
 RTC_PATTERN_SINE2D, RTC_PATTERN_TEXTURE_MAP, RTC_PATTERN_CUBE_MAP) = 0, 1, 2, 3, 4, 5, 6, 7
RTC_UV_CHECKERS, RTC_UV_ALIGN_CHECK, RTC_UV_IMAGE = 1, 2, 3
RTC_MAP_SPHERICAL, RTC_MAP_PLANAR, RTC_MAP_CYLINDRICAL = 1, 2, 3
RTC_LIGHT_POINT, RTC_LIGHT_RECT = 0, 1
RTC_JITTER_CONSTANT, RTC_JITTER_HASHED, RTC_JITTER_SEQUENCE = 0, 2, 3
RTC_MAX_DEPTH = 255        # accepted by rtc_render / rtc_ctx_render (above RTC_STACK_DEPTH_BASE: a scene kernel with a longer stack)
RTC_STACK_DEPTH_BASE = 8   # ... and by rtc_color_at

FP = C.POINTER(C.c_float)


class rtc_uv_pattern(C.Structure):
    _fields_ = [("kind", C.c_int32), ("width", C.c_float), ("height", C.c_float), ("colors", (C.c_float * 3) * 5),
                ("image_width", C.c_uint32), ("image_height", C.c_uint32), ("image_rgb", C.POINTER(C.c_float))]


class rtc_pattern(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_float * 3), ("b", C.c_float * 3), ("inv", C.c_float * 16),
                ("uv_mapping", C.c_int32), ("n_uv", C.c_uint32), ("uv", C.POINTER(rtc_uv_pattern))]


class rtc_material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("ambient", C.c_float), ("diffuse", C.c_float),
                ("specular", C.c_float), ("shininess", C.c_float), ("reflective", C.c_float),
                ("transparency", C.c_float), ("refractive_index", C.c_float), ("pattern", rtc_pattern)]


class rtc_object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("casts_shadow", C.c_int32), ("closed", C.c_int32),
                ("min_y", C.c_float), ("max_y", C.c_float), ("inv", C.c_float * 16),
                ("material", rtc_material), ("p1", C.c_float * 3), ("p2", C.c_float * 3), ("p3", C.c_float * 3)]


class rtc_light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("intensity", C.c_float * 3), ("position", C.c_float * 4),
                ("corner", C.c_float * 4), ("u_vec", C.c_float * 4), ("v_vec", C.c_float * 4),
                ("u_steps", C.c_int32), ("v_steps", C.c_int32), ("jitter_mode", C.c_int32),
                ("jitter_const", C.c_float), ("jitter_seed", C.c_uint32), ("jitter_seq_len", C.c_uint32),
                ("jitter_seq", C.c_float * 16)]


class rtc_group(C.Structure):
    _fields_ = [("first_object", C.c_uint32), ("n_objects", C.c_uint32), ("bounds_min", C.c_float * 3),
                ("bounds_max", C.c_float * 3)]


class rtc_scene(C.Structure):
    _fields_ = [("n_objects", C.c_uint32), ("objects", C.POINTER(rtc_object)),
                ("light", C.POINTER(rtc_light)), ("n_groups", C.c_uint32), ("groups", C.POINTER(rtc_group))]


class rtc_camera(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("field_of_view", C.c_float),
                ("half_width", C.c_float), ("half_height", C.c_float), ("pixel_size", C.c_float),
                ("inv", C.c_float * 16)]


class rtc_partition(C.Structure):
    _fields_ = [("band_rows", C.c_uint32), ("n_parts", C.c_uint32), ("part", C.c_uint32)]


class rtc_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("shaded_hits", C.c_uint64), ("pixels", C.c_uint64),
                ("kernel_ms", C.c_float), ("launches", C.c_uint32), ("rows", C.c_uint32),
                ("culled_shadow_rays", C.c_uint64), ("flags", C.c_uint32), ("gather_ms", C.c_float)]


RTC_STATS_JIT_FALLBACK = 1


class rtc_opts(C.Structure):
    _fields_ = [("devices", C.POINTER(C.c_int32)), ("n_devices", C.c_uint32), ("band_rows", C.c_uint32),
                ("quantize", C.c_int32), ("out_on_device", C.c_int32)]


# name -> (restype, argtypes); every symbol declared in include/rtc.h
SIGNATURES = {
    "rtc_translation": (None, [C.c_float] * 3 + [FP]),
    "rtc_scaling": (None, [C.c_float] * 3 + [FP]),
    "rtc_rotation_x": (None, [C.c_float, FP]),
    "rtc_rotation_y": (None, [C.c_float, FP]),
    "rtc_rotation_z": (None, [C.c_float, FP]),
    "rtc_shearing": (None, [C.c_float] * 6 + [FP]),
    "rtc_view_transform": (None, [FP, FP, FP, FP]),
    "rtc_mat_mul": (None, [FP, FP, FP]),
    "rtc_mat_vec": (None, [FP, FP, FP]),
    "rtc_mat_transpose": (None, [FP, C.c_int, FP]),
    "rtc_mat_determinant": (C.c_float, [FP, C.c_int]),
    "rtc_mat_submatrix": (None, [FP, C.c_int, C.c_int, C.c_int, FP]),
    "rtc_mat_minor": (C.c_float, [FP, C.c_int, C.c_int, C.c_int]),
    "rtc_mat_cofactor": (C.c_float, [FP, C.c_int, C.c_int, C.c_int]),
    "rtc_mat_inverse": (C.c_int, [FP, C.c_int, FP]),
    "rtc_magnitude": (C.c_float, [FP]),
    "rtc_norm": (None, [FP, FP]),
    "rtc_dot": (C.c_float, [FP, FP]),
    "rtc_cross": (None, [FP, FP, FP]),
    "rtc_reflect": (None, [FP, FP, FP]),
    "rtc_material_default": (None, [C.POINTER(rtc_material)]),
    "rtc_pattern_init": (C.c_int, [C.POINTER(rtc_pattern), C.c_int32, FP, FP, FP]),
    "rtc_texture_map_init": (C.c_int, [C.POINTER(rtc_pattern), C.c_int32, C.POINTER(rtc_uv_pattern), C.c_uint32, FP]),
    "rtc_canvas_from_ppm": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_void_p)]),
    "rtc_object_init": (C.c_int, [C.POINTER(rtc_object), C.c_int32, FP, C.POINTER(rtc_material)]),
    "rtc_bounds_empty": (None, [FP, FP]),
    "rtc_bounds_add": (None, [FP, FP, FP, FP]),
    "rtc_bounds_contains": (C.c_int32, [FP, FP, FP, FP]),
    "rtc_bounds_transform": (None, [FP, FP, FP, FP, FP]),
    "rtc_bounds_split": (None, [FP, FP, FP, FP, FP, FP]),
    "rtc_shape_bounds": (C.c_int, [C.c_int32, C.c_float, C.c_float, FP, FP, FP]),
    "rtc_triangle_bounds": (None, [FP, FP, FP, FP, FP, FP]),
    "rtc_triangle_fields": (None, [FP, FP, FP, FP, FP, FP]),
    "rtc_point_light": (None, [FP, FP, C.POINTER(rtc_light)]),
    "rtc_rectangle_light": (C.c_int, [FP, FP, FP, C.c_int32, FP, C.c_int32, C.c_int32, C.c_float, C.c_uint32,
                                      C.POINTER(rtc_light)]),
    "rtc_camera_new": (C.c_int, [C.c_uint32, C.c_uint32, C.c_float, FP, C.POINTER(rtc_camera)]),
    "rtc_ray_for_pixel": (None, [C.POINTER(rtc_camera), C.c_uint32, C.c_uint32, FP, FP]),
    "rtc_render": (C.c_int, [C.POINTER(rtc_scene), C.POINTER(rtc_camera), C.c_int32, C.c_int32, FP,
                             C.POINTER(rtc_stats)]),
    "rtc_render_ex": (C.c_int, [C.POINTER(rtc_scene), C.POINTER(rtc_camera), C.c_int32, C.POINTER(rtc_opts), C.c_void_p,
                                C.POINTER(rtc_stats)]),
    "rtc_render_release": (None, []),
    "rtc_host_alloc": (C.c_void_p, [C.c_size_t]),
    "rtc_host_free": (None, [C.c_void_p]),
    "rtc_scene_validate": (C.c_int, [C.POINTER(rtc_scene), C.POINTER(rtc_camera)]),
    "rtc_ctx_create": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p)]),
    "rtc_ctx_destroy": (None, [C.c_void_p]),
    "rtc_ctx_set_scene": (C.c_int, [C.c_void_p, C.POINTER(rtc_scene), C.POINTER(rtc_camera)]),
    "rtc_partition_rows": (C.c_uint32, [C.c_uint32, C.POINTER(rtc_partition)]),
    "rtc_ctx_render": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(rtc_partition), C.c_void_p, C.c_void_p]),
    "rtc_ctx_stats": (C.c_int, [C.c_void_p, C.POINTER(rtc_stats)]),
    "rtc_ctx_kernel_name": (C.c_char_p, [C.c_void_p]),
    "rtc_ctx_jit_status": (C.c_char_p, [C.c_void_p]),
    "rtc_ctx_kernel_id": (C.c_char_p, [C.c_void_p]),
    "rtc_ctx_quantize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rtc_ppm_max_bytes": (C.c_uint64, [C.c_uint32, C.c_uint32]),
    "rtc_ctx_to_ppm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
                                 C.POINTER(C.c_uint64), C.c_void_p]),
    "rtc_color_at": (C.c_int, [C.POINTER(rtc_scene), FP, FP, C.c_uint32, C.c_int32, C.c_int32, FP]),
    "rtc_intensity_at": (C.c_int, [C.POINTER(rtc_scene), FP, C.c_uint32, C.c_int32, FP]),
    "rtc_is_shadowed": (C.c_int, [C.POINTER(rtc_scene), FP, FP, C.c_uint32, C.c_int32, C.POINTER(C.c_int32)]),
    "rtc_point_on_light": (C.c_int, [C.POINTER(rtc_light), C.POINTER(C.c_int32), C.c_uint32, C.c_int32, FP]),
    "rtc_light_set_jitter_sequence": (C.c_int, [C.POINTER(rtc_light), FP, C.c_uint32]),
    "rtc_local_intersect": (C.c_int, [C.POINTER(rtc_object), FP, FP, C.c_uint32, C.c_int32, FP, C.POINTER(C.c_int32)]),
    "rtc_normal_at": (C.c_int, [C.POINTER(rtc_object), FP, C.c_uint32, C.c_int32, FP]),
    "rtc_pattern_color_at": (C.c_int, [C.POINTER(rtc_pattern), C.POINTER(rtc_object), FP, C.c_uint32, C.c_int32, FP]),
    "rtc_powf": (C.c_int, [FP, FP, C.c_uint32, C.c_int32, FP]),
    "rtc_cosf": (C.c_int, [FP, C.c_uint32, C.c_int32, FP]),
    "rtc_atan2f": (C.c_int, [FP, FP, C.c_uint32, C.c_int32, FP]),
    "rtc_acosf": (C.c_int, [FP, C.c_uint32, C.c_int32, FP]),
    "rtc_to_ppm": (C.c_int, [FP, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "rtc_free": (None, [C.c_void_p]),
    "rtc_last_error": (C.c_char_p, []),
    "rtc_abi_version": (C.c_int32, []),
    "rtc_device_count": (C.c_int32, []),
}
# diagnostic export, not in rtc.h: host compile of the device powf restatement
EXTRA = {"rtc_powf_host": (None, [FP, FP, C.c_uint32, FP]),
         "rtc_cosf_host": (None, [FP, C.c_uint32, FP]),
         "rtc_atan2f_host": (None, [FP, FP, C.c_uint32, FP]),
         "rtc_acosf_host": (None, [FP, C.c_uint32, FP]),
         # device self-test of the range-checked exact sqrt/divide cores against sqrtf and '/'
         "rtc_selftest_fastmath": (C.c_int, [FP, C.c_uint32, C.c_int32, C.POINTER(C.c_uint32)]),
         # 1 in librtc_amd_dev.so (built with -DRTC_DEV_SWITCHES), 0 in the library that ships
         "rtc_dev_switches": (C.c_int32, []),
         # the level-by-level renderer's counters after a context's last frame (tools / tests)
         "rtc_ctx_wavefront_counters": (C.c_uint32, [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]),
         # the block-list feedback's two host functions as rtc_ctx_render uses them (tests/test_block_lists.py)
         "rtc_diag_refine_block_list": (C.c_uint32, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_double,
                                                     C.c_double, C.c_double, C.POINTER(C.c_uint32), C.c_uint32]),
         "rtc_diag_simulate_dispatch": (C.c_double, [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32])}

_lib = None
_loaded = {}  # path -> CDLL
DEV_LIB_PATH = os.path.join(HERE, "librtc_amd_dev.so")


class RtcError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("rtc status %d: %s" % (status, message))
        self.status = status


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (same SONAME as
    /opt/rocm's).  If librtc_amd.so pulls in the system copy first and torch is imported afterwards,
    torch finds a runtime it was not built against and reports no GPU.  So, when torch is installed but
    not imported yet, load ITS runtime first (without importing torch): both then share it, whatever
    the import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load(path):
    """The library at `path`, with every signature declared (loaded once per path)."""
    path = os.path.abspath(path)
    if path not in _loaded:
        if not os.path.exists(path):
            raise ImportError(
                "%s is missing: build it with `python -m ray_tracer_challenge_amd.build%s` "
                "(there is no CPU/Python fallback for the render path)" % (path, " --dev" if path.endswith("_dev.so") else ""))
        _preload_torch_hip_runtime()
        L = C.CDLL(path)
        for name, (res, args) in list(SIGNATURES.items()) + list(EXTRA.items()):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _loaded[path] = L
    return _loaded[path]


def lib():
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
    return _lib


class use_library:
    """with use_library(path): ... -- every call of this package goes to that library inside the block (tests: the
    development build next to the one that ships).  Contexts remember the library that made them (Renderer)."""

    def __init__(self, path):
        self.path, self.prev = path, None

    def __enter__(self):
        global _lib
        self.prev = lib()
        _lib = load(self.path)
        return _lib

    def __exit__(self, *exc):
        global _lib
        try:
            _lib.rtc_render_release()  # what the one-call seam keeps between calls belongs to the library left behind
        finally:
            _lib = self.prev
        return False


def check(status, library=None):
    if status != RTC_OK:
        raise RtcError(status, (library or lib()).rtc_last_error().decode(errors="replace"))
