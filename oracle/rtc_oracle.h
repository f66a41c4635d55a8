/*
 * rtc_oracle.h -- C API of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C++ restatement of the per-pixel
 * render path of garfieldnate/ray_tracer_challenge (Rust, f32).  It exists so
 * that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
 * check (and time) the HIP path against the reference's arithmetic.  Nothing
 * in the product path (ray_tracer_challenge_amd/, include/) may include, link
 * or call anything in this directory.
 *
 * Parity status: the reference cannot be compiled here (no Rust toolchain), so
 * the oracle is pinned by the reference's own known-answer unit tests,
 * transcribed as data under tests/golden/ (see tests/test_oracle_*.py).
 *
 * All citations are file:line in /root/reference/lib/src.
 */
#ifndef RTC_ORACLE_H
#define RTC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RTCO_SPHERE = 0, RTCO_PLANE = 1, RTCO_CUBE = 2, RTCO_CYLINDER = 3, RTCO_CONE = 4,
       RTCO_TRIANGLE = 5,        /* shape/triangle.rs */
       RTCO_SMOOTH_TRIANGLE = 6, /* shape/smooth_triangle.rs: intersects through its inner Triangle (:37-39), so
                                    in World::intersect the hit object -- and the normal -- are the flat triangle's */
       RTCO_TEST_SHAPE = 100 /* shape/test_shape.rs: no hits, local normal (2x,3y,4z) */ };
enum { RTCO_LIGHT_POINT = 0, RTCO_LIGHT_RECT = 1 };
/* jitter sources for RectangleLight (light/rectangle_light.rs:44-47, test/utils.rs:15-24) */
enum { RTCO_JITTER_CONSTANT = 0, RTCO_JITTER_CYCLE = 1, RTCO_JITTER_HASHED = 2 };

/* pattern/{stripes,gradient,rings,checkers,sine_2d}.rs; RTCO_PATTERN_TEST is pattern/pattern.rs:66-89
 * (test double: the pattern-space point as the colour) */
enum { RTCO_PATTERN_NONE = 0, RTCO_PATTERN_STRIPES = 1, RTCO_PATTERN_GRADIENT = 2, RTCO_PATTERN_RINGS = 3,
       RTCO_PATTERN_CHECKERS = 4, RTCO_PATTERN_SINE2D = 5,
       RTCO_PATTERN_TEXTURE_MAP = 6, /* pattern/uv.rs:62-89: one UV pattern through a UV mapping */
       RTCO_PATTERN_CUBE_MAP = 7,    /* pattern/uv.rs:211-262: six UV patterns (front, back, left, right, up, down) */
       RTCO_PATTERN_TEST = 100 };
/* pattern/uv.rs: UVCheckers :20-55, AlignCheck :125-167, UVImage :347-377; mappings :91-113, :180-198 */
enum { RTCO_UV_CHECKERS = 1, RTCO_UV_ALIGN_CHECK = 2, RTCO_UV_IMAGE = 3 };
enum { RTCO_MAP_SPHERICAL = 1, RTCO_MAP_PLANAR = 2, RTCO_MAP_CYLINDRICAL = 3 };
typedef struct rtco_uv_pattern {
    int32_t kind;
    float width, height;     /* UVCheckers */
    float colors[5][3];      /* UVCheckers: a, b; AlignCheck: main, ul, ur, bl, br */
    uint32_t image_width, image_height;
    const float* image_rgb;  /* UVImage: Canvas.data, image_height rows of image_width RGB f32 (copied) */
} rtco_uv_pattern;

/* `transform` is the FORWARD pattern->object matrix; the oracle inverts it (pattern.rs:52-54). */
typedef struct rtco_pattern {
    int32_t kind;
    float a[3], b[3];
    float transform[16];
    int32_t uv_mapping;         /* TEXTURE_MAP */
    int32_t n_uv;               /* 1 (TEXTURE_MAP) or 6 (CUBE_MAP) */
    const rtco_uv_pattern* uv;
} rtco_pattern;

/* material.rs:18-51 */
typedef struct rtco_material {
    float color[3];
    float ambient, diffuse, specular, shininess;
    float reflective, transparency, refractive_index;
    rtco_pattern pattern;
} rtco_material;

/* shape/base_shape.rs:13-20 + cylinder.rs:14-19.  `transform` is the FORWARD
 * object->world matrix (row-major); the oracle inverts it itself. */
typedef struct rtco_shape {
    int32_t kind;
    int32_t casts_shadow;
    int32_t closed;
    float min_y, max_y;
    float transform[16];
    rtco_material material;
    float p1[4], p2[4], p3[4]; /* Triangle::new(p1, p2, p3), triangle.rs:19-33 */
    float n1[4], n2[4], n3[4]; /* SmoothTriangle::new(.., n1, n2, n3), smooth_triangle.rs:17-26 */
} rtco_shape;

/* light/point_light.rs:7-10, light/rectangle_light.rs:12-31.  u_vec/v_vec are
 * the FULL edge vectors as passed to RectangleLight::new. */
typedef struct rtco_light {
    int32_t kind;
    float intensity[3];
    float position[4]; /* point light position */
    float corner[4];
    float u_vec[4];
    int32_t u_steps;
    float v_vec[4];
    int32_t v_steps;
    int32_t jitter_mode;
    float jitter_const;
    uint32_t jitter_seed;
    const float* jitter_seq; /* RTCO_JITTER_CYCLE: copied at world creation */
    int32_t jitter_seq_len;
} rtco_light;

/* camera.rs:8-21 after Camera::new */
typedef struct rtco_camera {
    uint32_t width, height;
    float field_of_view;
    float half_width, half_height, pixel_size;
    float transform_inverse[16];
} rtco_camera;

/* world.rs:165-182 */
typedef struct rtco_comps {
    float distance;
    int32_t object;
    float point[4], eye[4], reflectv[4], normal[4], over_point[4], under_point[4];
    int32_t inside;
    float n1, n2;
} rtco_comps;

typedef struct rtco_world rtco_world;

/* ---- tuple.rs / ray.rs ---- */
float rtco_magnitude(const float v[4]);
void rtco_norm(const float v[4], float out[4]);
float rtco_dot(const float a[4], const float b[4]);
void rtco_cross(const float a[4], const float b[4], float out[4]);
void rtco_reflect(const float in[4], const float n[4], float out[4]);
void rtco_position(const float o[4], const float d[4], float t, float out[4]);

/* ---- matrix.rs (n = 2,3,4; row-major n*n) ---- */
void rtco_mat_mul(const float a[16], const float b[16], float out[16]);
void rtco_mat_vec(const float a[16], const float v[4], float out[4]);
void rtco_mat_transpose(const float* a, int n, float* out);
float rtco_mat_determinant(const float* a, int n);
void rtco_mat_submatrix(const float* a, int n, int row, int col, float* out);
float rtco_mat_minor(const float* a, int n, int row, int col);
float rtco_mat_cofactor(const float* a, int n, int row, int col);
void rtco_mat_inverse(const float* a, int n, float* out);

/* ---- transformations.rs ---- */
void rtco_translation(float x, float y, float z, float out[16]);
void rtco_scaling(float x, float y, float z, float out[16]);
void rtco_rotation_x(float r, float out[16]);
void rtco_rotation_y(float r, float out[16]);
void rtco_rotation_z(float r, float out[16]);
void rtco_shearing(float xy, float xz, float yx, float yz, float zx, float zy, float out[16]);
void rtco_view_transform(const float from[4], const float to[4], const float up[4], float out[16]);

/* ---- camera.rs ---- */
void rtco_camera_new(uint32_t w, uint32_t h, float fov, const float transform[16], rtco_camera* out);
void rtco_ray_for_pixel(const rtco_camera* c, uint32_t x, uint32_t y, float o[4], float d[4]);

/* ---- shapes ---- */
int rtco_local_intersect(const rtco_shape* s, const float o[4], const float d[4], float ts[4]);
void rtco_local_normal_at(const rtco_shape* s, const float p[4], float out[4]);
/* local_intersect with the (u, v) each intersection carries (intersection.rs:8-9) */
int rtco_local_intersect_uv(const rtco_shape* s, const float o[4], const float d[4], float ts[4], float us[4],
                            float vs[4]);
/* Shape::normal_at(point, hit) for a hit constructed by hand with (u, v) -- the only way the reference ever
 * reaches SmoothTriangle::local_norm_at (smooth_triangle.rs:41-43) */
void rtco_normal_at_uv(const rtco_shape* s, const float world_point[4], float u, float v, float out[4]);
/* Triangle's derived fields e1, e2, normal (triangle.rs:20-22) */
void rtco_triangle_fields(const rtco_shape* s, float e1[4], float e2[4], float normal[4]);
int rtco_shape_intersect(const rtco_shape* s, const float o[4], const float d[4], float ts[4],
                         float obj_o[4], float obj_d[4]);
void rtco_normal_at(const rtco_shape* s, const float world_point[4], float out[4]);
int rtco_aabb_intersection(const float o[4], const float d[4], const float mn[4], const float mx[4],
                           float out_t[2]);
/* intersection.rs:30-35; returns index or -1 */
int rtco_hit(const float* ts, int n);

/* ---- world.rs / lights ---- */
rtco_world* rtco_world_new(const rtco_shape* shapes, int n, const rtco_light* light);
void rtco_world_free(rtco_world* w);

/* ---- shape/group.rs + bounding_box.rs.  Nodes (leaf shapes and GroupShapes) live in one process-wide
 * arena and are addressed by integer ids; the functions restate the reference methods of the same name. ---- */
int rtco_node_shape(const rtco_shape* s);
int rtco_node_group(void);                                        /* GroupShape::new(), group.rs:19-21 */
int rtco_node_group_with_children(const int* children, int n);    /* group.rs:23-27 (nothing re-baked) */
void rtco_node_add_child(int group, int child);                   /* group.rs:39-44 */
void rtco_node_set_transformation(int node, const float t[16]);   /* group.rs:101-114 / base_shape.rs:56-60 */
void rtco_node_set_material(int node, const rtco_material* m);    /* group.rs:96-100 */
void rtco_node_divide(int node, uint32_t threshold);              /* group.rs:157-172 */
int rtco_node_is_group(int node);
int rtco_node_children(int node, int* out, int cap);
void rtco_node_transformation(int node, float out[16]);
float rtco_node_shininess(int node);
void rtco_node_bounding_box(int node, float mn[4], float mx[4]);               /* group.rs:138-151 (cached) */
void rtco_node_parent_space_bounding_box(int node, float mn[4], float mx[4]);  /* shape.rs:162-164 / group.rs:153-155 */
int rtco_node_intersect(int node, const float o[4], const float d[4], float* ts, int* leaf_nodes, int cap);
void rtco_node_world_to_object(int node, const float p[4], float out[4]);
void rtco_node_normal_at(int node, const float p[4], float out[4]);
rtco_world* rtco_world_new_nodes(const int* roots, int n, const rtco_light* light);
void rtco_bbox_empty(float mn[4], float mx[4]);
void rtco_bbox_add_point(float mn[4], float mx[4], const float p[4]);
void rtco_bbox_add(float mn[4], float mx[4], const float omn[4], const float omx[4]);
int rtco_bbox_contains_point(const float mn[4], const float mx[4], const float p[4]);
int rtco_bbox_contains(const float mn[4], const float mx[4], const float omn[4], const float omx[4]);
void rtco_bbox_transform(const float mn[4], const float mx[4], const float m[16], float omn[4], float omx[4]);
void rtco_bbox_split(const float mn[4], const float mx[4], float lmn[4], float lmx[4], float rmn[4], float rmx[4]);
void rtco_shape_bounding_box(const rtco_shape* s, int parent_space, float mn[4], float mx[4]);
void rtco_world_set_pixel(rtco_world* w, uint32_t pixel_index); /* key for hashed jitter */
uint64_t rtco_world_ray_count(const rtco_world* w);
void rtco_world_shape_inverse(const rtco_world* w, int i, float inv[16], float inv_t[16]);
void rtco_light_info(const rtco_world* w, float position[4], float u_vec[4], float v_vec[4], int* cells);
int rtco_intersect(rtco_world* w, const float o[4], const float d[4], float* ts, int* objs, int cap);
void rtco_color_at(rtco_world* w, const float o[4], const float d[4], int depth, float out[3]);
int rtco_is_shadowed(rtco_world* w, const float light_pos[4], const float p[4]);
float rtco_intensity_at(rtco_world* w, const float p[4]);
void rtco_point_on_light(rtco_world* w, int u, int v, float out[4]);
void rtco_precompute(rtco_world* w, const float o[4], const float d[4], int hit, const float* ts,
                     const int* objs, int n, rtco_comps* out);
void rtco_shade_hit(rtco_world* w, const rtco_comps* c, int depth, float out[3]);
void rtco_reflected_color(rtco_world* w, const rtco_comps* c, int depth, float out[3]);
void rtco_refracted_color(rtco_world* w, const rtco_comps* c, int depth, float out[3]);
float rtco_schlick(const rtco_comps* c);
/* light/phong_lighting.rs:12-63; light taken from the world */
void rtco_phong(rtco_world* w, const rtco_material* m, const float p[4], const float eye[4],
                const float n[4], float light_intensity, float out[3]);
/* same with the lit object given (phong_lighting.rs:13), for patterned materials */
void rtco_phong_on(rtco_world* w, const rtco_shape* object, const float p[4], const float eye[4],
                   const float n[4], float light_intensity, float out[3]);
/* pattern.rs:12 color_at_world / :15-19 color_at_object */
void rtco_pattern_color_at_world(const rtco_pattern* pat, const float p[4], float out[3]);
/* pattern/uv.rs pieces, as the reference's tests call them */
void rtco_uv_color_at(const rtco_uv_pattern* uv, float u, float v, float out[3]);
void rtco_point_to_uv(int32_t mapping, const float p[4], float uv[2]);
int rtco_face_from_point(const float p[4]); /* 0 front, 1 back, 2 left, 3 right, 4 up, 5 down (uv.rs:169-177) */
void rtco_cube_uv(int face, const float p[4], float uv[2]);
/* canvas.rs:120-197 canvas_from_ppm: returns 0 and a malloc'd w*h*3 f32 image (free with rtco_free), or the
 * index of the ParseError variant + 1 (1 IoError, 2 IncorrectFormat, 3 ParseIntError, 4 MalformedDimensionHeader) */
int rtco_canvas_from_ppm(const char* text, uint64_t len, uint32_t* w, uint32_t* h, float** rgb);
void rtco_pattern_color_at_object(const rtco_pattern* pat, const rtco_shape* object, const float world_point[4],
                                  float out[3]);

/* ---- camera.rs:76-91 render + canvas.rs ---- */
/* out_rgb: w*h*3 f32 row-major, fully written (last row/column black).
 * threads<=1: the reference's serial loop.  Returns rays traced. */
uint64_t rtco_render(rtco_world* w, const rtco_camera* c, int depth, int threads, float* out_rgb);
/* renders rows [y0,y1) only (still skipping the last row/column), for bounded CPU timing */
uint64_t rtco_render_rows(rtco_world* w, const rtco_camera* c, int depth, int threads,
                          uint32_t y0, uint32_t y1, float* out_rgb);
uint8_t rtco_scale_color(float c);
void rtco_quantize(const float* rgb, uint64_t n, uint8_t* out);
/* canvas.rs:58-96; returns malloc'd NUL-terminated text, length in *len; free with rtco_free */
char* rtco_to_ppm(const float* rgb, uint32_t w, uint32_t h, uint64_t* len);
void rtco_free(void* p);

/* jitter spec shared (by specification, not by code) with the HIP kernel */
uint32_t rtco_jitter_hash(uint32_t seed, uint32_t pixel, uint32_t path, uint32_t cell, uint32_t draw);
float rtco_jitter_value(uint32_t h);

#ifdef __cplusplus
}
#endif
#endif
