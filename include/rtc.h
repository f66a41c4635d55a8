/*
 * rtc.h -- C ABI of the MI355X-native render path (librtc_amd.so).
 *
 * This is the drop-in boundary for ONE hot path of
 * garfieldnate/ray_tracer_challenge: the body of
 *
 *     pub fn render(&self, world: World, reflection_recursion_depth: i16) -> Canvas
 *                                                  (lib/src/camera.rs:76-91)
 *
 * and everything it calls per pixel (ray_for_pixel, World::color_at / intersect /
 * shade_hit / is_shadowed / reflected_color / refracted_color, the Sphere /
 * Plane / Cube / Cylinder / Cone intersectors, phong_lighting with the
 * procedural patterns of the pattern module).  The reference has no
 * FFI of its own; INTEGRATION.md shows the `extern "C"` block a maintainer
 * would add to camera.rs to bind these entry points.
 *
 * Plain C: POD structs, pointers and sizes only.  No C++ or torch types cross
 * this boundary.  The library owns all device memory it allocates; the caller
 * owns every buffer it passes in.  All matrices are row-major 4x4 f32.
 *
 * File:line citations are relative to /root/reference/lib/src.
 */
#ifndef RTC_H
#define RTC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTC_ABI_VERSION 8
/* reflection_recursion_depth (camera.rs:76: any i16; reference default 5, constants.rs:4; its author renders
 * reflect_refract at 20).  Accepted: 0 .. RTC_MAX_DEPTH.  The kernels keep one frame per suspended shade_hit
 * (world.rs:62-86); up to RTC_STACK_DEPTH_BASE levels every kernel has them, above that the scene's kernel is compiled
 * once more, on first use, with a stack of 16 / 32 / ... levels in per-lane scratch (a second or so; cached like every
 * scene kernel).  Beyond RTC_MAX_DEPTH the call is refused -- the reference would be recursing that deep on its own
 * call stack.  rtc_color_at (a batched test entry point) stops at RTC_STACK_DEPTH_BASE. */
#define RTC_MAX_DEPTH 255
#define RTC_STACK_DEPTH_BASE 8

typedef enum rtc_status {
    RTC_OK = 0,
    RTC_ERR_INVALID_ARG = -1, /* null pointer, zero size, depth out of range ...        */
    RTC_ERR_UNSUPPORTED = -2, /* unknown shape kind, projective transform, closure jitter */
    RTC_ERR_NO_LIGHT = -3,    /* world.rs:66 "World light should be set"                 */
    RTC_ERR_DEVICE = -4,      /* HIP runtime error; text via rtc_last_error()            */
    RTC_ERR_NO_DEVICE = -5    /* no gfx950 device visible: there is NO CPU fallback      */
} rtc_status;

/* shape/{sphere,plane,cube,cylinder,cone,triangle}.rs.  A SmoothTriangle is passed as RTC_TRIANGLE: its
 * local_intersect hands out intersections whose object is the inner flat Triangle (smooth_triangle.rs:37-39), so
 * inside Camera::render it is shaded exactly like one. */
enum { RTC_SPHERE = 0, RTC_PLANE = 1, RTC_CUBE = 2, RTC_CYLINDER = 3, RTC_CONE = 4, RTC_TRIANGLE = 5 };
/* pattern/{stripes,gradient,rings,checkers,sine_2d}.rs; NONE = Material.pattern is None (material.rs:50) */
enum { RTC_PATTERN_NONE = 0, RTC_PATTERN_STRIPES = 1, RTC_PATTERN_GRADIENT = 2, RTC_PATTERN_RINGS = 3,
       RTC_PATTERN_CHECKERS = 4, RTC_PATTERN_SINE2D = 5,
       RTC_PATTERN_TEXTURE_MAP = 6, /* pattern/uv.rs:62-89 TextureMap: one UV pattern through a UV mapping   */
       RTC_PATTERN_CUBE_MAP = 7     /* pattern/uv.rs:200-262 CubicMap: six UV patterns, one per cube face     */ };
/* pattern/uv.rs: UVCheckers :20-55, AlignCheck :125-167, UVImage :347-377 */
enum { RTC_UV_CHECKERS = 1, RTC_UV_ALIGN_CHECK = 2, RTC_UV_IMAGE = 3 };
/* pattern/uv.rs: SphericalMap :91-105, PlanarMap :180-186, CylindricalMap :188-198 */
enum { RTC_MAP_SPHERICAL = 1, RTC_MAP_PLANAR = 2, RTC_MAP_CYLINDRICAL = 3 };
/* light/{point_light,rectangle_light}.rs */
enum { RTC_LIGHT_POINT = 0, RTC_LIGHT_RECT = 1 };
/* RectangleLight jitter source.  The reference takes an arbitrary closure
 * (rectangle_light.rs:27,44-47); a device cannot call one, so two pinned
 * sources are offered: the constant of test/utils.rs:15-17, and a counter-based
 * hash (DESIGN.md "Jitter") standing in for thread_rng(). */
enum { RTC_JITTER_CONSTANT = 0, RTC_JITTER_HASHED = 2,
       /* test/utils.rs:19-24 hardcoded_jitter: a short list of values handed out in a cycle.  The cycle is STATE that
        * the reference carries from call to call (a RefCell inside the closure), serial across everything a light is
        * asked: only the entry points that stand for ONE call on a freshly built light accept it -- rtc_intensity_at
        * (every point is answered as by a new light: draw k of the call is value k mod n, two draws per cell in
        * rectangle_light.rs:76-88's order) and rtc_point_on_light; rtc_render / rtc_ctx_set_scene / rtc_color_at refuse. */
       RTC_JITTER_SEQUENCE = 3 };
#define RTC_JITTER_SEQUENCE_MAX 16

/* A boxed Pattern (pattern/pattern.rs:8-26) flattened: the two colours every pattern is built from and
 * BasePattern.t_inverse (pattern.rs:33,52-54).  Gradient and Sine2D keep `distance = b - a`
 * (gradient.rs:17, sine_2d.rs:17); the library forms it from a and b with the same single subtraction.
 * Build with rtc_pattern_init().  (uv.rs texture maps are not on this path.) */
/* One boxed UVPattern (pattern/uv.rs:14-16). */
typedef struct rtc_uv_pattern {
    int32_t kind;           /* RTC_UV_* */
    float width, height;    /* UVCheckers */
    float colors[5][3];     /* UVCheckers: a, b.  AlignCheck: main, ul, ur, bl, br */
    uint32_t image_width;   /* UVImage: its Canvas (canvas.rs:6-10) as image_height rows of image_width RGB f32,  */
    uint32_t image_height;  /* i.e. Canvas.data[y][x]; host memory, copied into HBM by rtc_ctx_set_scene          */
    const float* image_rgb;
} rtc_uv_pattern;

typedef struct rtc_pattern {
    int32_t kind; /* RTC_PATTERN_* */
    float a[3];
    float b[3];
    float inv[16];
    int32_t uv_mapping;       /* TEXTURE_MAP: RTC_MAP_*                                                         */
    uint32_t n_uv;            /* TEXTURE_MAP: 1.  CUBE_MAP: 6, in CubicMap::new's order front, back, left, right, */
    const rtc_uv_pattern* uv; /* up, down (uv.rs:207-231).  Read during rtc_ctx_set_scene / the batched calls.    */
} rtc_pattern;

/* material.rs:18-51 */
typedef struct rtc_material {
    float color[3];
    float ambient;
    float diffuse;
    float specular;
    float shininess;
    float reflective;
    float transparency;
    float refractive_index;
    rtc_pattern pattern; /* kind RTC_PATTERN_NONE: use `color` (phong_lighting.rs:24-27) */
} rtc_material;

/* One entry of World.objects (world.rs:19), flattened.  `inv` is exactly
 * BaseShape.t_inverse (shape/base_shape.rs:17,58); the inverse-transpose is
 * its transpose and is not stored.  Object order is World.objects order: hit
 * tie-breaking depends on it (world.rs:58, intersection.rs:30-35). */
typedef struct rtc_object {
    int32_t kind;         /* RTC_SPHERE ... */
    int32_t casts_shadow; /* BaseShape.casts_shadow, base_shape.rs:14 */
    int32_t closed;       /* Cylinder.closed / Cone.closed, cylinder.rs:18, cone.rs:16 */
    float min_y;          /* Cylinder/Cone.minimum_y (ignored for other kinds) */
    float max_y;          /* Cylinder/Cone.maximum_y */
    float inv[16];
    rtc_material material;
    float p1[3], p2[3], p3[3]; /* Triangle.p1..p3 (triangle.rs:11-13); e1, e2 and the normal are re-derived by
                                  the library exactly as Triangle::new does (:20-22).  Ignored for other kinds. */
} rtc_object;

/* light/point_light.rs:7-10 and light/rectangle_light.rs:12-31 AFTER
 * RectangleLight::new: u_vec/v_vec are the per-cell vectors (already divided by
 * the step counts, :51-52) and `position` is the rectangle centre (:57) or the
 * point light's position.  Build with rtc_point_light()/rtc_rectangle_light(). */
typedef struct rtc_light {
    int32_t kind;
    float intensity[3];
    float position[4];
    float corner[4];
    float u_vec[4];
    float v_vec[4];
    int32_t u_steps;
    int32_t v_steps;
    int32_t jitter_mode;
    float jitter_const;
    uint32_t jitter_seed;
    uint32_t jitter_seq_len;                      /* RTC_JITTER_SEQUENCE: 1 .. RTC_JITTER_SEQUENCE_MAX values ... */
    float jitter_seq[RTC_JITTER_SEQUENCE_MAX];    /* ... set by rtc_light_set_jitter_sequence()                   */
} rtc_light;

/* One GroupShape (shape/group.rs:12-16) of the flattened world.  add_child / set_transformation bake a
 * group's transform into its children (group.rs:39-44,101-114) and Shape::intersect on a group does not
 * transform the ray (:115-117), so a tree of groups flattens to its leaf shapes -- listed in `objects` in
 * depth-first order, which is the order the reference's child loops push intersections in -- plus, per
 * group, the contiguous run of leaves under it and the bounding box that gates them (:119-133).
 * Groups are listed in pre-order (a group before the groups nested inside it); runs nest properly. */
typedef struct rtc_group {
    uint32_t first_object;
    uint32_t n_objects;   /* 0: an empty group (never hit; ignored) */
    float bounds_min[3];  /* GroupShape::bounding_box(), group.rs:138-151 (world space) */
    float bounds_max[3];
} rtc_group;

/* world.rs:18-21 */
typedef struct rtc_scene {
    uint32_t n_objects;
    const rtc_object* objects;
    const rtc_light* light; /* NULL -> RTC_ERR_NO_LIGHT */
    uint32_t n_groups;      /* 0: World.objects is a flat list of shapes */
    const rtc_group* groups;
} rtc_scene;

/* camera.rs:8-21 after Camera::new.  Build with rtc_camera_new(). */
typedef struct rtc_camera {
    uint32_t width;
    uint32_t height;
    float field_of_view;
    float half_width;
    float half_height;
    float pixel_size;
    float inv[16]; /* transform_inverse */
} rtc_camera;

/* Row partition of one image over several devices (one process per GPU).
 * The image is cut into bands of `band_rows` rows; band b belongs to part
 * (b mod n_parts).  A part's rows are stored compactly, band after band. */
typedef struct rtc_partition {
    uint32_t band_rows; /* 0 -> 64 */
    uint32_t n_parts;   /* 0 -> 1  */
    uint32_t part;
} rtc_partition;

typedef struct rtc_stats {
    uint64_t rays;        /* World::intersect evaluations of the last launch (primary, shadow, reflect, refract) */
    uint64_t shaded_hits; /* shade_hit evaluations of the last launch (world.rs:62)                           */
    uint64_t pixels;      /* traced pixels: (w-1)*(h-1) restricted to the rows rendered                        */
    float kernel_ms;      /* mean HIP-event time of the render kernel over `launches`                          */
    uint32_t launches;    /* render launches since the previous rtc_ctx_stats call                             */
    uint32_t rows;        /* rows written to the output buffer                                                 */
    uint64_t culled_shadow_rays; /* of `rays`: area-light shadow rays whose answer ("lit") followed from the
                                    conservative light-cone cull, i.e. that tested no object       */
    uint32_t flags;       /* RTC_STATS_*                                                                       */
    float gather_ms;      /* rtc_render_ex: wall time from the first render launch to the last row's arrival in `out`
                             minus nothing -- i.e. the whole render + transport pipeline of the call (0 elsewhere)  */
} rtc_stats;
/* rtc_stats.flags: the scene's kernel is an ahead-of-time instantiation although the specialisation policy asked for a
 * scene-compiled one (hiprtc failed; same image, several times slower on area-light scenes).  rtc_ctx_jit_status()
 * holds the compiler's message. */
#define RTC_STATS_JIT_FALLBACK 1u

/* Options of rtc_render_ex (SURVEY.md 8(b) `opts`).  Zero-initialise for the defaults. */
typedef struct rtc_opts {
    const int32_t* devices; /* the GPUs to split the image over (64-row bands dealt round-robin, camera.rs:80-85 has no   */
    uint32_t n_devices;     /* order dependence between pixels); NULL / 0: device 0.  A device may be listed twice.      */
    uint32_t band_rows;     /* 0 -> 64 */
    int32_t quantize;       /* 0: `out` receives width*height*3 f32 (the Canvas);  1: width*height*3 u8 -- every channel */
                            /* through scale_color (canvas.rs:39-43), i.e. the numbers Canvas::to_ppm prints             */
    int32_t out_on_device;  /* 0: `out` is host memory (pinned memory from rtc_host_alloc is written without staging);   */
                            /* 1: `out` is device memory on devices[0]: peers' bands travel GPU-to-GPU (xGMI)            */
} rtc_opts;

typedef struct rtc_ctx rtc_ctx;

/* ------------------------------------------------------------------------
 * Host-side scene math.  Restates matrix.rs / transformations.rs /
 * tuple.rs / camera.rs:23-74 operation for operation so that the flattened
 * scene handed to the kernel is bit-identical to what the Rust structs hold.
 * Pure host code; usable without a GPU.
 * ---------------------------------------------------------------------- */
void rtc_translation(float x, float y, float z, float out[16]);                 /* transformations.rs:4-6   */
void rtc_scaling(float x, float y, float z, float out[16]);                     /* :8-10  */
void rtc_rotation_x(float radians, float out[16]);                              /* :12-21 */
void rtc_rotation_y(float radians, float out[16]);                              /* :23-32 */
void rtc_rotation_z(float radians, float out[16]);                              /* :34-43 */
void rtc_shearing(float xy, float xz, float yx, float yz, float zx, float zy, float out[16]); /* :46-53 */
void rtc_view_transform(const float from[4], const float to[4], const float up[4], float out[16]); /* :57-68 */
void rtc_mat_mul(const float a[16], const float b[16], float out[16]);          /* matrix.rs:86-103  */
void rtc_mat_vec(const float a[16], const float v[4], float out[4]);            /* matrix.rs:73-84   */
void rtc_mat_transpose(const float* a, int n, float* out);                      /* matrix.rs:134-143 */
float rtc_mat_determinant(const float* a, int n);                               /* matrix.rs:145-160 */
void rtc_mat_submatrix(const float* a, int n, int row, int col, float* out);    /* matrix.rs:163-182 */
float rtc_mat_minor(const float* a, int n, int row, int col);                   /* matrix.rs:194-196 */
float rtc_mat_cofactor(const float* a, int n, int row, int col);                /* matrix.rs:184-192 */
rtc_status rtc_mat_inverse(const float* a, int n, float* out);                  /* matrix.rs:201-212 */
float rtc_magnitude(const float v[4]);                                          /* tuple.rs:29-33 */
void rtc_norm(const float v[4], float out[4]);                                  /* tuple.rs:34-43 */
float rtc_dot(const float a[4], const float b[4]);                              /* tuple.rs:44-46 */
void rtc_cross(const float a[4], const float b[4], float out[4]);               /* tuple.rs:47-55 */
void rtc_reflect(const float in[4], const float normal[4], float out[4]);       /* ray.rs:42-44   */

/* Material::default(), material.rs:53-57 (no pattern) */
void rtc_material_default(rtc_material* out);
/* Stripes/Gradient/Rings/Checkers/Sine2D::new(a, b) followed by set_transformation(transform):
 * stores transform.inverse() (pattern.rs:52-54).  transform NULL = identity. */
rtc_status rtc_pattern_init(rtc_pattern* out, int32_t kind, const float a[3], const float b[3],
                            const float transform[16]);
/* TextureMap::new(uv_pattern, uv_mapping) (uv.rs:68-76; uv_mapping = RTC_MAP_*, n_uv = 1) or CubicMap::new(front,
 * back, left, right, up, down) (uv.rs:207-231; uv_mapping = 0, n_uv = 6), then set_transformation(transform).
 * `uv` is borrowed, not copied. */
rtc_status rtc_texture_map_init(rtc_pattern* out, int32_t uv_mapping, const rtc_uv_pattern* uv, uint32_t n_uv,
                                const float transform[16]);
/* canvas_from_ppm (canvas.rs:120-197): parses P3 text into a malloc'd width*height*3 f32 image (free with
 * rtc_free).  Errors: RTC_ERR_INVALID_ARG with the ParseError variant's name in rtc_last_error(). */
rtc_status rtc_canvas_from_ppm(const char* text, uint64_t len, uint32_t* width, uint32_t* height, float** out_rgb);
/* Shape::build(transform, material): stores transform.inverse() (base_shape.rs:56-60).
 * Cylinder / Cone bounds default to -inf/+inf, open (cylinder.rs:34-43, cone.rs:33-42). */
rtc_status rtc_object_init(rtc_object* out, int32_t kind, const float transform[16], const rtc_material* m);
/* bounding_box.rs, operation for operation (f32::min / max ignore a NaN operand; points carry w = 1).
 * Host-side helpers for building rtc_group records the way GroupShape::bounding_box / divide do. */
void rtc_bounds_empty(float mn[4], float mx[4]);                                                  /* :13-20  */
void rtc_bounds_add(float mn[4], float mx[4], const float other_mn[4], const float other_mx[4]);  /* :37-50  */
int32_t rtc_bounds_contains(const float mn[4], const float mx[4], const float other_mn[4],
                            const float other_mx[4]);                                             /* :52-60  */
void rtc_bounds_transform(const float mn[4], const float mx[4], const float m[16], float out_mn[4],
                          float out_mx[4]);                                                       /* :62-79  */
void rtc_bounds_split(const float mn[4], const float mx[4], float left_mn[4], float left_mx[4],
                      float right_mn[4], float right_mx[4]);                                      /* :85-113 */
/* Shape::bounding_box (sphere.rs:75-80, plane.rs:61-66, cube.rs:82-87, cylinder.rs:74-79, cone.rs:75-85);
 * with transform != NULL, Shape::parent_space_bounding_box (shape.rs:162-164) for that transformation(). */
rtc_status rtc_shape_bounds(int32_t kind, float min_y, float max_y, const float transform[16], float mn[4],
                            float mx[4]);
/* the same for a Triangle (triangle.rs:74-80) */
void rtc_triangle_bounds(const float p1[3], const float p2[3], const float p3[3], const float transform[16],
                         float mn[4], float mx[4]);
/* Triangle::new's derived fields (triangle.rs:20-22): e1 = p2 - p1, e2 = p3 - p1, normal = e2.cross(e1).norm() */
void rtc_triangle_fields(const float p1[3], const float p2[3], const float p3[3], float e1[3], float e2[3],
                         float normal[3]);
void rtc_point_light(const float position[4], const float intensity[3], rtc_light* out);   /* point_light.rs:12-19 */
rtc_status rtc_rectangle_light(const float intensity[3], const float corner[4], const float u_vec[4],
                               int32_t u_steps, const float v_vec[4], int32_t v_steps, int32_t jitter_mode,
                               float jitter_const, uint32_t jitter_seed, rtc_light* out); /* rectangle_light.rs:33-58 */
/* hardcoded_jitter(values) of test/utils.rs:19-24 for a light built with jitter mode RTC_JITTER_SEQUENCE (see the enum) */
rtc_status rtc_light_set_jitter_sequence(rtc_light* light, const float* values, uint32_t n);
rtc_status rtc_camera_new(uint32_t width, uint32_t height, float field_of_view, const float transform[16],
                          rtc_camera* out);                                               /* camera.rs:23-56 */
void rtc_ray_for_pixel(const rtc_camera* c, uint32_t x, uint32_t y, float origin[4], float direction[4]); /* camera.rs:60-74 */

/* Runs the checks and the flattening rtc_ctx_set_scene performs (shape / pattern kinds, affine transforms, group
 * nesting, light and camera sanity), without touching a device: RTC_OK, or the status rtc_ctx_set_scene would
 * return with its message in rtc_last_error().  camera may be NULL. */
rtc_status rtc_scene_validate(const rtc_scene* scene, const rtc_camera* camera);

/* ------------------------------------------------------------------------
 * Device path.  Every function below needs a gfx950 GPU and fails with
 * RTC_ERR_NO_DEVICE otherwise.
 * ---------------------------------------------------------------------- */

/* Camera::render (camera.rs:76-91) in one call: uploads the scene, renders the
 * whole image on device `device`, copies it back.  out_rgb: caller-owned host
 * buffer of width*height*3 f32, row-major [y][x][rgb], fully written -- the
 * last row and last column stay black exactly as in the reference
 * (camera.rs:80-81).  stats may be NULL. */
rtc_status rtc_render(const rtc_scene* scene, const rtc_camera* camera, int32_t depth, int32_t device,
                      float* out_rgb, rtc_stats* stats);

/* The same with options: several devices (the image's bands are dealt round-robin over opts->devices, each device
 * renders its bands while the previous ones are on their way to `out`), bytes instead of floats, device-resident
 * output.  Contexts, device buffers, pinned staging memory and compiled kernels are kept per device between calls
 * (rtc_render_release drops them), so a second frame costs the kernel plus the transfer.  stats: rays / shaded hits /
 * pixels summed over the devices, kernel_ms = the largest per-device sum of kernel times, gather_ms = wall time of
 * the render + transfer pipeline.  rtc_render(scene, camera, depth, device, out, stats) is rtc_render_ex with
 * opts = {devices = &device, n_devices = 1}.
 * Which kernel: the reference renders one frame per process (camera.rs:76), and compiling a scene's own kernel takes 0.5 - 2 s
 * where the frame it speeds up takes milliseconds.  These calls therefore render a scene with the ahead-of-time kernels the first time a
 * process sees it -- unless its compiled kernel is already in the disk cache (<library dir>/jit_cache) -- and compile it when the same
 * scene is rendered again, for that process and, through the cache, for every later one.  Same frames either way.
 * RTC_AMD_SPECIALIZE=1 compiles at first sight, =0 never; the persistent contexts below (rtc_ctx_set_scene) always compile at once. */
rtc_status rtc_render_ex(const rtc_scene* scene, const rtc_camera* camera, int32_t depth, const rtc_opts* opts,
                         void* out, rtc_stats* stats);
/* Frees everything rtc_render / rtc_render_ex keep between calls (all devices). */
void rtc_render_release(void);
/* Page-locked host memory (hipHostMalloc): an `out` buffer allocated here is filled by DMA straight from the device. */
void* rtc_host_alloc(size_t bytes);
void rtc_host_free(void* p);

/* Persistent context: scene resident in HBM, output left on the device.
 * Ordering contract: a context serves ONE stream at a time.  rtc_ctx_render is asynchronous; launches on the same
 * stream queue up behind each other, but rendering from one context on two streams concurrently is a race on the
 * context's counters.  rtc_ctx_set_scene waits for every launch the context has issued before it replaces the scene
 * (it synchronises the device), so it may be called right after an asynchronous render. */
rtc_status rtc_ctx_create(int32_t device, rtc_ctx** out);
void rtc_ctx_destroy(rtc_ctx* ctx);
/* Flattens the scene to structure-of-arrays records and uploads it.  Called again with the scene that is resident: nothing
 * happens; with records that are resident and another camera (an animation's usual frame): nothing is uploaded; with a scene
 * of the same frame size: what the frames before measured stays the schedule's starting point. */
rtc_status rtc_ctx_set_scene(rtc_ctx* ctx, const rtc_scene* scene, const rtc_camera* camera);
/* Rows this partition produces (sum of its bands' heights). */
uint32_t rtc_partition_rows(uint32_t height, const rtc_partition* part);
/* Launches the render kernel on `stream` (a hipStream_t; NULL = default
 * stream) and returns without synchronising.  d_out_rgb: DEVICE pointer to
 * rtc_partition_rows()*width*3 f32.  part may be NULL (whole image).
 * "Returns without synchronising" has one exception per scene, partition and depth: the context schedules a frame by what
 * the frame before it measured (which blocks of pixels start first, how many lanes trace a pixel of each),
 * and the second -- for block lists also the third -- call reads those measurements back, which waits for the device.
 * Every frame traces every ray and returns the same values.  RTC_AMD_BLOCK_FEEDBACK=0: every frame like the first. */
rtc_status rtc_ctx_render(rtc_ctx* ctx, int32_t depth, const rtc_partition* part, void* d_out_rgb, void* stream);
/* Waits for every rtc_ctx_render issued on this context so far and reports the
 * last launch's counters plus the mean kernel time since the previous call. */
rtc_status rtc_ctx_stats(rtc_ctx* ctx, rtc_stats* out);
/* Name of the kernel rtc_ctx_render launches for the current scene: an ahead-of-time instantiation
 * ("render_kernel<4,simple>") or a scene-specialised one compiled at rtc_ctx_set_scene
 * ("render_kernel_spec[...]"; env RTC_AMD_SPECIALIZE=0|1 overrides the size-based default). */
const char* rtc_ctx_kernel_name(rtc_ctx* ctx);
/* "" when the current scene's kernel is what the specialisation policy asked for; otherwise why it is not (the hiprtc
 * failure), in which case rtc_stats.flags carries RTC_STATS_JIT_FALLBACK and a warning went to stderr once (silence:
 * RTC_AMD_QUIET=1).  The kernel source is embedded in the library: no file beside librtc_amd.so is read at run time.
 * Compiled kernels are cached in <library dir>/jit_cache, or RTC_AMD_JIT_CACHE=<dir> (0: memory only). */
const char* rtc_ctx_jit_status(rtc_ctx* ctx);
/* Names the CODE the current scene is rendered with -- "spec_<hash of kernel source, compile options and compiler
 * version>.<checksum of the compiled code object>" or "aot_<hash of the kernel source>" -- so that a measurement taken of one kernel (the profiles/ *_pmc.json summaries)
 * is never quoted for another: bench.py prints instruction counts and HBM traffic only from a summary whose id matches. */
const char* rtc_ctx_kernel_id(rtc_ctx* ctx);
/* canvas.rs:39-43 scale_color on the device: n f32 channel values -> n bytes
 * ((c*255).min(255).max(0) as u8).  Both pointers are device pointers. */
rtc_status rtc_ctx_quantize(rtc_ctx* ctx, const void* d_rgb, uint64_t n, void* d_out_u8, void* stream);

/* Canvas::to_ppm (canvas.rs:58-96) on the device: formats an f32 image that is already in HBM
 * (d_rgb: height*width*3 f32) as P3 text into d_text (DEVICE buffer of at least
 * rtc_ppm_max_bytes(width, height) bytes), byte-for-byte what the reference builds, including the
 * 70-column wrapping.  Synchronises `stream`; *out_len = bytes written. */
uint64_t rtc_ppm_max_bytes(uint32_t width, uint32_t height);
rtc_status rtc_ctx_to_ppm(rtc_ctx* ctx, const void* d_rgb, uint32_t width, uint32_t height, void* d_text,
                          uint64_t capacity, uint64_t* out_len, void* stream);

/* Batched World::color_at (world.rs:88-101) for caller-supplied rays; ray i
 * uses pixel index i as its jitter key.  Host buffers: origins/directions
 * n*4 f32, out n*3 f32. */
rtc_status rtc_color_at(const rtc_scene* scene, const float* origins, const float* directions, uint32_t n,
                        int32_t depth, int32_t device, float* out_rgb);
/* Batched Light::intensity_at (light.rs:10) for n world points (n*4 f32). */
rtc_status rtc_intensity_at(const rtc_scene* scene, const float* points, uint32_t n, int32_t device,
                            float* out);
/* Batched RectangleLight::point_on_light (rectangle_light.rs:60-66) on the device: n (u, v) cell pairs (n*2 int32), each
 * answered as by a freshly built light (a sequence jitter's first two values; a hashed one's cell key with pixel 0, path
 * 1); out n*4 f32 points. */
rtc_status rtc_point_on_light(const rtc_light* light, const int32_t* cells_uv, uint32_t n, int32_t device, float* out);
/* Batched World::is_shadowed (world.rs:104-119): light_positions, points n*4 f32; out n int32 (0/1). */
rtc_status rtc_is_shadowed(const rtc_scene* scene, const float* light_positions, const float* points, uint32_t n,
                           int32_t device, int32_t* out);
/* Batched Shape::local_intersect (shape.rs:50, e.g. cone.rs:52-57) on the device: n object-space rays
 * (n*4 f32 each) against `object`'s kind/bounds; out_t receives up to 4 distances per ray in push
 * order (n*4 f32), out_count the number pushed (n int32). */
rtc_status rtc_local_intersect(const rtc_object* object, const float* origins, const float* directions, uint32_t n,
                               int32_t device, float* out_t, int32_t* out_count);
/* Batched Shape::normal_at (shape.rs:72-154: object space, local_norm_at, back to world, normalise)
 * for n world points (n*4 f32); out n*4 f32 vectors. */
rtc_status rtc_normal_at(const rtc_object* object, const float* world_points, uint32_t n, int32_t device,
                         float* out);
/* Batched Pattern::color_at_object (pattern.rs:15-19) for n world points (n*4 f32) on `object`
 * (NULL: an untransformed shape); out n*3 f32. */
rtc_status rtc_pattern_color_at(const rtc_pattern* pattern, const rtc_object* object, const float* world_points,
                                uint32_t n, int32_t device, float* out_rgb);
/* f32::powf / f32::cos as the reference's Linux build computes them (phong_lighting.rs:56,
 * pattern/sine_2d.rs:40), evaluated on the device; host buffers of n f32. */
rtc_status rtc_powf(const float* x, const float* y, uint32_t n, int32_t device, float* out);
rtc_status rtc_cosf(const float* x, uint32_t n, int32_t device, float* out);
/* f32::atan2 / f32::acos (pattern/uv.rs:108,101) likewise */
rtc_status rtc_atan2f(const float* y, const float* x, uint32_t n, int32_t device, float* out);
rtc_status rtc_acosf(const float* x, uint32_t n, int32_t device, float* out);

/* Canvas::to_ppm (canvas.rs:58-96) on an f32 image already on the host:
 * returns a malloc'd buffer (free with rtc_free) holding the P3 text. */
rtc_status rtc_to_ppm(const float* rgb, uint32_t width, uint32_t height, char** out_text, uint64_t* out_len);
void rtc_free(void* p);

/* Thread-local text of the last error returned on this thread. */
const char* rtc_last_error(void);
int32_t rtc_abi_version(void);
/* Number of usable gfx950 devices (0 if none / no HIP runtime). */
int32_t rtc_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* RTC_H */
