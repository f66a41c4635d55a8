// rtc_host.cpp -- host-side scene math of librtc_amd.so (no GPU needed).
//
// The kernel consumes *inverse* transforms, the camera's pixel size and the
// area light's per-cell vectors.  The reference computes all of those once per
// scene on the CPU with a particular sequence of f32 operations (recursive
// cofactor expansion, divide-by-determinant, libm tanf/sinf/cosf ...).  To be a
// drop-in, the flattened scene must hold the same bits, so this file restates
// that arithmetic -- with fixed-size templates instead of Vec<Vec<f32>>.
//
// Compiled with -ffp-contract=off: rustc does not contract a*b+c.
// Citations: file:line under /root/reference/lib/src.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rtc_internal.h"

namespace rtc {

thread_local std::string g_last_error;

rtc_status fail(rtc_status code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// ---- matrix.rs:145-196: determinant by cofactor expansion along row 0 -------
template <int N>
struct Det {
    static float of(const float* a) {
        float det = 0.0f;  // `let mut det = 0.0; det += cofactor * data[0][col]` (matrix.rs:151-158)
        for (int col = 0; col < N; col++) {
            float sub[(N - 1) * (N - 1)];
            submatrix<N>(a, 0, col, sub);
            float minor = Det<N - 1>::of(sub);
            float cof = (col % 2 == 0) ? minor : -minor;
            det += cof * a[col];
        }
        return det;
    }
};
template <>
struct Det<2> {
    static float of(const float* a) { return a[0] * a[3] - a[1] * a[2]; }  // matrix.rs:147-148
};

template <int N>
float minor_of(const float* a, int row, int col) {
    float sub[(N - 1) * (N - 1)];
    submatrix<N>(a, row, col, sub);
    return Det<N - 1>::of(sub);
}
template <>
float minor_of<2>(const float* a, int row, int col) {
    return a[(1 - row) * 2 + (1 - col)];  // 1x1 remainder (never used by the reference; kept total)
}

template <int N>
float cofactor_of(const float* a, int row, int col) {  // matrix.rs:184-192
    float m = minor_of<N>(a, row, col);
    return ((row + col) % 2 == 0) ? m : -m;
}

template <int N>
void inverse_of(const float* a, float* out) {  // matrix.rs:201-212
    float det = Det<N>::of(a);
    for (int row = 0; row < N; row++)
        for (int col = 0; col < N; col++) out[col * N + row] = cofactor_of<N>(a, row, col) / det;
}

float determinant(const float* a, int n) {
    switch (n) {
        case 2: return Det<2>::of(a);
        case 3: return Det<3>::of(a);
        default: return Det<4>::of(a);
    }
}

void inverse4(const float a[16], float out[16]) { inverse_of<4>(a, out); }

void mat_mul4(const float a[16], const float b[16], float out[16]) {  // matrix.rs:86-103
    float r[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r[i * 4 + j] = a[i * 4 + 0] * b[0 * 4 + j] + a[i * 4 + 1] * b[1 * 4 + j] + a[i * 4 + 2] * b[2 * 4 + j] +
                           a[i * 4 + 3] * b[3 * 4 + j];
    std::memcpy(out, r, sizeof(r));
}

void mat_vec4(const float a[16], const float v[4], float out[4]) {  // matrix.rs:73-84
    float r[4];
    for (int i = 0; i < 4; i++) r[i] = a[i * 4] * v[0] + a[i * 4 + 1] * v[1] + a[i * 4 + 2] * v[2] + a[i * 4 + 3] * v[3];
    std::memcpy(out, r, sizeof(r));
}

static void set16(float out[16], float a, float b, float c, float d, float e, float f, float g, float h, float i,
                  float j, float k, float l, float m, float n, float o, float p) {
    const float v[16] = {a, b, c, d, e, f, g, h, i, j, k, l, m, n, o, p};
    std::memcpy(out, v, sizeof(v));
}

float magnitude4(const float v[4]) {  // tuple.rs:29-33
    return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
}
void norm4(const float v[4], float out[4]) {  // tuple.rs:34-43 (w untouched)
    float m = magnitude4(v);
    float r[4] = {v[0] / m, v[1] / m, v[2] / m, v[3]};
    std::memcpy(out, r, sizeof(r));
}
void cross4(const float a[4], const float b[4], float out[4]) {  // tuple.rs:47-55
    float r[4] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0], 0.0f};
    std::memcpy(out, r, sizeof(r));
}

bool is_affine(const float m[16]) { return m[12] == 0.0f && m[13] == 0.0f && m[14] == 0.0f && m[15] == 1.0f; }

}  // namespace rtc

using namespace rtc;

extern "C" {

const char* rtc_last_error(void) { return g_last_error.c_str(); }
int32_t rtc_abi_version(void) { return RTC_ABI_VERSION; }

void rtc_translation(float x, float y, float z, float out[16]) { set16(out, 1, 0, 0, x, 0, 1, 0, y, 0, 0, 1, z, 0, 0, 0, 1); }
void rtc_scaling(float x, float y, float z, float out[16]) { set16(out, x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1); }
// f32::cos / f32::sin lower to libm cosf / sinf on Linux (transformations.rs:13-14)
void rtc_rotation_x(float r, float out[16]) {
    float c = cosf(r), s = sinf(r);
    set16(out, 1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1);
}
void rtc_rotation_y(float r, float out[16]) {
    float c = cosf(r), s = sinf(r);
    set16(out, c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1);
}
void rtc_rotation_z(float r, float out[16]) {
    float c = cosf(r), s = sinf(r);
    set16(out, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1);
}
void rtc_shearing(float xy, float xz, float yx, float yz, float zx, float zy, float out[16]) {
    set16(out, 1, xy, xz, 0, yx, 1, yz, 0, zx, zy, 1, 0, 0, 0, 0, 1);
}
void rtc_view_transform(const float from[4], const float to[4], const float up[4], float out[16]) {
    // transformations.rs:57-68
    float diff[4] = {to[0] - from[0], to[1] - from[1], to[2] - from[2], to[3] - from[3]};
    float forward[4], upn[4], left[4], true_up[4];
    norm4(diff, forward);
    norm4(up, upn);
    cross4(forward, upn, left);
    cross4(left, forward, true_up);
    float orientation[16], tr[16];
    set16(orientation, left[0], left[1], left[2], 0, true_up[0], true_up[1], true_up[2], 0, -forward[0], -forward[1],
          -forward[2], 0, 0, 0, 0, 1);
    rtc_translation(-from[0], -from[1], -from[2], tr);
    mat_mul4(orientation, tr, out);
}
void rtc_mat_mul(const float a[16], const float b[16], float out[16]) { mat_mul4(a, b, out); }
void rtc_mat_vec(const float a[16], const float v[4], float out[4]) { mat_vec4(a, v, out); }
void rtc_mat_transpose(const float* a, int n, float* out) {
    float r[16];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) r[j * n + i] = a[i * n + j];
    std::memcpy(out, r, sizeof(float) * n * n);
}
float rtc_mat_determinant(const float* a, int n) { return determinant(a, n); }
void rtc_mat_submatrix(const float* a, int n, int row, int col, float* out) {
    if (n == 4) submatrix<4>(a, row, col, out);
    else if (n == 3) submatrix<3>(a, row, col, out);
    else submatrix<2>(a, row, col, out);
}
float rtc_mat_minor(const float* a, int n, int row, int col) {
    return n == 4 ? minor_of<4>(a, row, col) : n == 3 ? minor_of<3>(a, row, col) : minor_of<2>(a, row, col);
}
float rtc_mat_cofactor(const float* a, int n, int row, int col) {
    return n == 4 ? cofactor_of<4>(a, row, col) : n == 3 ? cofactor_of<3>(a, row, col) : cofactor_of<2>(a, row, col);
}
rtc_status rtc_mat_inverse(const float* a, int n, float* out) {
    if (!a || !out || n < 2 || n > 4) return fail(RTC_ERR_INVALID_ARG, "rtc_mat_inverse: n must be 2..4");
    float r[16];
    if (n == 4) inverse_of<4>(a, r);
    else if (n == 3) inverse_of<3>(a, r);
    else inverse_of<2>(a, r);
    std::memcpy(out, r, sizeof(float) * n * n);
    return RTC_OK;
}
float rtc_magnitude(const float v[4]) { return magnitude4(v); }
void rtc_norm(const float v[4], float out[4]) { norm4(v, out); }
float rtc_dot(const float a[4], const float b[4]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + (a[3] * b[3]); }
void rtc_cross(const float a[4], const float b[4], float out[4]) { cross4(a, b, out); }
void rtc_reflect(const float in[4], const float n[4], float out[4]) {
    // ray.rs:43: -(normal * 2.0 * in.dot(normal) - in)
    float d = rtc_dot(in, n);
    float r[4];
    for (int i = 0; i < 4; i++) r[i] = -(n[i] * 2.0f * d - in[i]);
    std::memcpy(out, r, sizeof(r));
}

void rtc_material_default(rtc_material* m) {  // material.rs:20-47
    m->color[0] = m->color[1] = m->color[2] = 1.0f;
    m->ambient = 0.1f;
    m->diffuse = 0.9f;
    m->specular = 0.9f;
    m->shininess = 200.0f;
    m->reflective = 0.0f;
    m->transparency = 0.0f;
    m->refractive_index = 1.0f;
    std::memset(&m->pattern, 0, sizeof(m->pattern));  // pattern: None
    m->pattern.kind = RTC_PATTERN_NONE;
    for (int i = 0; i < 4; i++) m->pattern.inv[i * 5] = 1.0f;
}

rtc_status rtc_pattern_init(rtc_pattern* out, int32_t kind, const float a[3], const float b[3],
                            const float transform[16]) {
    if (!out || !a || !b) return fail(RTC_ERR_INVALID_ARG, "rtc_pattern_init: null argument");
    if (kind < RTC_PATTERN_STRIPES || kind > RTC_PATTERN_SINE2D)
        return fail(RTC_ERR_UNSUPPORTED, "rtc_pattern_init: unknown pattern kind %d", kind);
    std::memset(out, 0, sizeof(*out));
    out->kind = kind;
    std::memcpy(out->a, a, sizeof(float) * 3);
    std::memcpy(out->b, b, sizeof(float) * 3);
    if (transform) {
        inverse4(transform, out->inv);  // pattern.rs:52-54
    } else {  // BasePattern::default(): the identity (pattern.rs:31-34)
        for (int i = 0; i < 4; i++) out->inv[i * 5] = 1.0f;
    }
    return RTC_OK;
}

rtc_status rtc_texture_map_init(rtc_pattern* out, int32_t uv_mapping, const rtc_uv_pattern* uv, uint32_t n_uv,
                                const float transform[16]) {
    if (!out || !uv) return fail(RTC_ERR_INVALID_ARG, "rtc_texture_map_init: null argument");
    const bool cube = n_uv == 6 && uv_mapping == 0;
    if (!cube && !(n_uv == 1 && uv_mapping >= RTC_MAP_SPHERICAL && uv_mapping <= RTC_MAP_CYLINDRICAL))
        return fail(RTC_ERR_UNSUPPORTED, "rtc_texture_map_init: want one UV pattern and a mapping (TextureMap) or six UV patterns "
                                         "and mapping 0 (CubicMap); got %u pattern(s), mapping %d", n_uv, uv_mapping);
    std::memset(out, 0, sizeof(*out));
    out->kind = cube ? RTC_PATTERN_CUBE_MAP : RTC_PATTERN_TEXTURE_MAP;
    out->uv_mapping = uv_mapping;
    out->n_uv = n_uv;
    out->uv = uv;
    if (transform) {
        inverse4(transform, out->inv);
    } else {
        for (int i = 0; i < 4; i++) out->inv[i * 5] = 1.0f;
    }
    return RTC_OK;
}

// canvas_from_ppm, canvas.rs:120-197 (clean_line :183-197): '#' lines and blank lines are dropped anywhere, then the
// magic "P3", "width height", the scale, and colour samples that may be split across lines any way they like; each
// complete triplet is one pixel (sample / scale as f32), filled row by row.
rtc_status rtc_canvas_from_ppm(const char* text, uint64_t len, uint32_t* width, uint32_t* height, float** out_rgb) {
    if (!text || !width || !height || !out_rgb) return fail(RTC_ERR_INVALID_ARG, "rtc_canvas_from_ppm: null argument");
    struct Line {
        const char* b;
        const char* e;
    };
    auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; };
    std::vector<Line> lines;
    for (uint64_t i = 0; i < len;) {
        uint64_t j = i;
        while (j < len && text[j] != '\n') j++;
        const char* b = text + i;
        const char* e = text + j;
        while (b < e && is_space(*b)) b++;
        while (e > b && is_space(e[-1])) e--;
        if (b < e && *b != '#') lines.push_back({b, e});
        i = j + 1;
    }
    // str::parse::<u32 / usize>: an optional '+', then decimal digits only
    auto parse_uint = [](const char* b, const char* e, uint64_t* out) {
        if (b < e && *b == '+') b++;
        if (b >= e) return false;
        uint64_t v = 0;
        for (; b < e; b++) {
            if (*b < '0' || *b > '9') return false;
            v = v * 10 + (uint64_t)(*b - '0');
            if (v > 0xffffffffull) return false;
        }
        *out = v;
        return true;
    };
    auto tokens = [&](const Line& l) {
        std::vector<Line> out;
        const char* p = l.b;
        while (p < l.e) {
            while (p < l.e && is_space(*p)) p++;
            const char* q = p;
            while (q < l.e && !is_space(*q)) q++;
            if (q > p) out.push_back({p, q});
            p = q;
        }
        return out;
    };
    if (lines.size() < 3) return fail(RTC_ERR_INVALID_ARG, "IoError: fewer than three header lines");  // the reference unwraps
    if (!(lines[0].e - lines[0].b == 2 && lines[0].b[0] == 'P' && lines[0].b[1] == '3'))
        return fail(RTC_ERR_INVALID_ARG, "IncorrectFormat: Incorrect magic number at line 1: expected P3, found %.*s",
                    (int)(lines[0].e - lines[0].b), lines[0].b);
    std::vector<Line> dims = tokens(lines[1]);
    if (dims.size() != 2)
        return fail(RTC_ERR_INVALID_ARG, "MalformedDimensionHeader: Expected width and height at line 2; found %.*s",
                    (int)(lines[1].e - lines[1].b), lines[1].b);
    uint64_t w = 0, h = 0, scale_u = 0;
    if (!parse_uint(dims[0].b, dims[0].e, &w) || !parse_uint(dims[1].b, dims[1].e, &h) || !parse_uint(lines[2].b, lines[2].e, &scale_u))
        return fail(RTC_ERR_INVALID_ARG, "ParseIntError: invalid digit found in string");
    const float scale = (float)(uint32_t)scale_u;
    // the header comes from an untrusted file: w * h * 3 * sizeof(f32) must not wrap, and an image the library's own
    // texel index (32 bits, pack_uv_pattern) cannot address is refused here rather than at rtc_ctx_set_scene
    size_t n = 0, n_bytes = 0;
    if (__builtin_mul_overflow((size_t)w, (size_t)h, &n) || n > 0xffffffffull || __builtin_mul_overflow(n, (size_t)3, &n) ||
        __builtin_mul_overflow(n, sizeof(float), &n_bytes))
        return fail(RTC_ERR_INVALID_ARG, "canvas of %llu x %llu pixels is too large", (unsigned long long)w, (unsigned long long)h);
    float* img = (float*)std::calloc(n ? n : 1, sizeof(float));
    if (!img) return fail(RTC_ERR_INVALID_ARG, "out of memory (%llu x %llu canvas)", (unsigned long long)w, (unsigned long long)h);
    uint32_t pending[3];
    int have = 0;
    size_t x = 0, y = 0;
    for (size_t li = 3; li < lines.size(); li++) {
        for (const Line& t : tokens(lines[li])) {
            uint64_t v;
            if (!parse_uint(t.b, t.e, &v)) {
                std::free(img);
                return fail(RTC_ERR_INVALID_ARG, "ParseIntError: invalid digit found in string");
            }
            pending[have++] = (uint32_t)v;
            if (have == 3) {
                have = 0;
                if (!(x < w && y < h)) {  // canvas.rs:27 write_pixel indexes data[y][x]: the reference panics here
                    std::free(img);
                    return fail(RTC_ERR_INVALID_ARG, "PixelOutOfBounds: more pixel data than the %llu x %llu header announces "
                                "(the reference panics in Canvas::write_pixel)", (unsigned long long)w, (unsigned long long)h);
                }
                float* px = img + (y * w + x) * 3;
                for (int k = 0; k < 3; k++) px[k] = (float)pending[k] / scale;
                if (++x >= w) {
                    x = 0;
                    y++;
                }
            }
        }
    }
    *width = (uint32_t)w;
    *height = (uint32_t)h;
    *out_rgb = img;
    return RTC_OK;
}

rtc_status rtc_object_init(rtc_object* out, int32_t kind, const float transform[16], const rtc_material* m) {
    if (!out || !transform) return fail(RTC_ERR_INVALID_ARG, "rtc_object_init: null argument");
    if (kind < RTC_SPHERE || kind > RTC_TRIANGLE) return fail(RTC_ERR_UNSUPPORTED, "rtc_object_init: unknown shape kind %d", kind);
    std::memset(out->p1, 0, sizeof(out->p1));
    std::memset(out->p2, 0, sizeof(out->p2));
    std::memset(out->p3, 0, sizeof(out->p3));
    out->kind = kind;
    out->casts_shadow = 1;  // base_shape.rs:31
    out->closed = 0;        // cylinder.rs:41
    out->min_y = -INFINITY; // cylinder.rs:39
    out->max_y = INFINITY;  // cylinder.rs:40
    inverse4(transform, out->inv);  // base_shape.rs:58
    if (m) out->material = *m;
    else rtc_material_default(&out->material);
    return RTC_OK;
}

// ---- bounding_box.rs ----
static void bounds_add_point(float mn[4], float mx[4], const float p[4]) {  // :37-45
    for (int k = 0; k < 3; k++) {
        mn[k] = fminf(mn[k], p[k]);
        mx[k] = fmaxf(mx[k], p[k]);
    }
}
void rtc_bounds_empty(float mn[4], float mx[4]) {  // :13-20
    for (int k = 0; k < 3; k++) {
        mn[k] = INFINITY;
        mx[k] = -INFINITY;
    }
    mn[3] = mx[3] = 1.0f;
}
void rtc_bounds_add(float mn[4], float mx[4], const float omn[4], const float omx[4]) {  // :47-50
    bounds_add_point(mn, mx, omn);
    bounds_add_point(mn, mx, omx);
}
static bool bounds_contains_point(const float mn[4], const float mx[4], const float p[4]) {  // :52-56
    for (int k = 0; k < 3; k++)
        if (!(p[k] >= mn[k] && p[k] <= mx[k])) return false;
    return true;
}
int32_t rtc_bounds_contains(const float mn[4], const float mx[4], const float omn[4], const float omx[4]) {  // :58-60
    return bounds_contains_point(mn, mx, omn) && bounds_contains_point(mn, mx, omx) ? 1 : 0;
}
void rtc_bounds_transform(const float mn[4], const float mx[4], const float m[16], float out_mn[4], float out_mx[4]) {
    // :62-79: the eight corners in the reference's order, each through the full 4x4 product
    float amn[4], amx[4];
    rtc_bounds_empty(amn, amx);
    for (int c = 0; c < 8; c++) {
        const float p[4] = {(c & 4) ? mx[0] : mn[0], (c & 2) ? mx[1] : mn[1], (c & 1) ? mx[2] : mn[2], 1.0f};
        float q[4];
        mat_vec4(m, p, q);
        bounds_add_point(amn, amx, q);
    }
    std::memcpy(out_mn, amn, sizeof(amn));
    std::memcpy(out_mx, amx, sizeof(amx));
}
void rtc_bounds_split(const float mn[4], const float mx[4], float lmn[4], float lmx[4], float rmn[4], float rmx[4]) {
    // :85-113
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    float greatest = fmaxf(fmaxf(dx, dy), dz);
    float x0 = mn[0], y0 = mn[1], z0 = mn[2];
    float x1 = mx[0], y1 = mx[1], z1 = mx[2];
    if (greatest == dx) {
        x0 = x0 + dx / 2.0f;
        x1 = x0;
    } else if (greatest == dy) {
        y0 = y0 + dy / 2.0f;
        y1 = y0;
    } else {
        z0 = z0 + dz / 2.0f;
        z1 = z0;
    }
    const float a[4] = {mn[0], mn[1], mn[2], 1.0f}, b[4] = {mx[0], mx[1], mx[2], 1.0f};
    const float mid_min[4] = {x0, y0, z0, 1.0f}, mid_max[4] = {x1, y1, z1, 1.0f};
    std::memcpy(lmn, a, sizeof(a));
    std::memcpy(lmx, mid_max, sizeof(a));
    std::memcpy(rmn, mid_min, sizeof(a));
    std::memcpy(rmx, b, sizeof(a));
}
rtc_status rtc_shape_bounds(int32_t kind, float min_y, float max_y, const float transform[16], float mn[4], float mx[4]) {
    if (!mn || !mx) return fail(RTC_ERR_INVALID_ARG, "rtc_shape_bounds: null argument");
    float a[4] = {-1.0f, -1.0f, -1.0f, 1.0f}, b[4] = {1.0f, 1.0f, 1.0f, 1.0f};  // sphere.rs:75-80, cube.rs:82-87
    switch (kind) {
        case RTC_SPHERE:
        case RTC_CUBE:
            break;
        case RTC_PLANE:  // plane.rs:61-66
            a[0] = a[2] = -INFINITY;
            b[0] = b[2] = INFINITY;
            a[1] = b[1] = 0.0f;
            break;
        case RTC_CYLINDER:  // cylinder.rs:74-79
            a[1] = min_y;
            b[1] = max_y;
            break;
        case RTC_CONE: {  // cone.rs:75-85
            float limit = fmaxf(fabsf(min_y), fabsf(max_y));
            a[0] = a[2] = -limit;
            b[0] = b[2] = limit;
            a[1] = min_y;
            b[1] = max_y;
            break;
        }
        default:
            return fail(RTC_ERR_UNSUPPORTED, "rtc_shape_bounds: unknown shape kind %d", kind);
    }
    if (transform) {
        rtc_bounds_transform(a, b, transform, mn, mx);  // shape.rs:162-164
    } else {
        std::memcpy(mn, a, sizeof(a));
        std::memcpy(mx, b, sizeof(b));
    }
    return RTC_OK;
}

void rtc_triangle_bounds(const float p1[3], const float p2[3], const float p3[3], const float transform[16], float mn[4],
                         float mx[4]) {
    float a[4], b[4];
    rtc_bounds_empty(a, b);  // triangle.rs:74-80
    const float* ps[3] = {p1, p2, p3};
    for (const float* p : ps) {
        const float q[4] = {p[0], p[1], p[2], 1.0f};
        bounds_add_point(a, b, q);
    }
    if (transform) {
        rtc_bounds_transform(a, b, transform, mn, mx);
    } else {
        std::memcpy(mn, a, sizeof(a));
        std::memcpy(mx, b, sizeof(b));
    }
}
void rtc_triangle_fields(const float p1[3], const float p2[3], const float p3[3], float e1[3], float e2[3], float normal[3]) {
    for (int k = 0; k < 3; k++) {
        e1[k] = p2[k] - p1[k];
        e2[k] = p3[k] - p1[k];
    }
    const float a[4] = {e2[0], e2[1], e2[2], 0.0f}, b[4] = {e1[0], e1[1], e1[2], 0.0f};
    float c[4], n[4];
    rtc_cross(a, b, c);
    rtc_norm(c, n);
    for (int k = 0; k < 3; k++) normal[k] = n[k];
}

void rtc_point_light(const float position[4], const float intensity[3], rtc_light* out) {
    std::memset(out, 0, sizeof(*out));
    out->kind = RTC_LIGHT_POINT;
    std::memcpy(out->position, position, sizeof(float) * 4);
    std::memcpy(out->intensity, intensity, sizeof(float) * 3);
    out->u_steps = out->v_steps = 1;
}

rtc_status rtc_rectangle_light(const float intensity[3], const float corner[4], const float u_vec[4], int32_t u_steps,
                               const float v_vec[4], int32_t v_steps, int32_t jitter_mode, float jitter_const,
                               uint32_t jitter_seed, rtc_light* out) {
    if (!out || !intensity || !corner || !u_vec || !v_vec) return fail(RTC_ERR_INVALID_ARG, "rtc_rectangle_light: null argument");
    if (u_steps <= 0 || v_steps <= 0) return fail(RTC_ERR_INVALID_ARG, "rtc_rectangle_light: steps must be positive");
    if (jitter_mode != RTC_JITTER_CONSTANT && jitter_mode != RTC_JITTER_HASHED && jitter_mode != RTC_JITTER_SEQUENCE)
        return fail(RTC_ERR_UNSUPPORTED, "rtc_rectangle_light: jitter mode %d cannot run on the device", jitter_mode);
    std::memset(out, 0, sizeof(*out));
    out->kind = RTC_LIGHT_RECT;
    std::memcpy(out->intensity, intensity, sizeof(float) * 3);
    std::memcpy(out->corner, corner, sizeof(float) * 4);
    for (int i = 0; i < 4; i++) {
        out->u_vec[i] = u_vec[i] / (float)u_steps;  // rectangle_light.rs:51
        out->v_vec[i] = v_vec[i] / (float)v_steps;  // :52
        // :57  corner + (u_vec / 2.) + (v_vec / 2.)
        out->position[i] = corner[i] + (u_vec[i] / 2.0f) + (v_vec[i] / 2.0f);
    }
    out->u_steps = u_steps;
    out->v_steps = v_steps;
    out->jitter_mode = jitter_mode;
    out->jitter_const = jitter_const;
    out->jitter_seed = jitter_seed;
    return RTC_OK;
}

// test/utils.rs:19-24 hardcoded_jitter
rtc_status rtc_light_set_jitter_sequence(rtc_light* light, const float* values, uint32_t n) {
    if (!light || !values) return fail(RTC_ERR_INVALID_ARG, "rtc_light_set_jitter_sequence: null argument");
    if (light->kind != RTC_LIGHT_RECT) return fail(RTC_ERR_INVALID_ARG, "rtc_light_set_jitter_sequence: not a rectangle light");
    if (n < 1 || n > RTC_JITTER_SEQUENCE_MAX)
        return fail(RTC_ERR_INVALID_ARG, "rtc_light_set_jitter_sequence: %u values (1 .. %d)", n, RTC_JITTER_SEQUENCE_MAX);
    light->jitter_mode = RTC_JITTER_SEQUENCE;
    light->jitter_seq_len = n;
    for (uint32_t i = 0; i < RTC_JITTER_SEQUENCE_MAX; i++) light->jitter_seq[i] = i < n ? values[i] : 0.0f;
    return RTC_OK;
}

rtc_status rtc_camera_new(uint32_t width, uint32_t height, float fov, const float transform[16], rtc_camera* out) {
    if (!out || !transform) return fail(RTC_ERR_INVALID_ARG, "rtc_camera_new: null argument");
    if (width == 0 || height == 0) return fail(RTC_ERR_INVALID_ARG, "rtc_camera_new: empty canvas");
    // camera.rs:35-46; f32::tan -> libm tanf
    float half_view = tanf(fov / 2.0f);
    float aspect = (float)width / (float)height;
    if (aspect >= 1.0f) {
        out->half_width = half_view;
        out->half_height = half_view / aspect;
    } else {
        out->half_width = half_view * aspect;
        out->half_height = half_view;
    }
    out->pixel_size = (out->half_width * 2.0f) / (float)width;
    out->width = width;
    out->height = height;
    out->field_of_view = fov;
    inverse4(transform, out->inv);
    return RTC_OK;
}

void rtc_ray_for_pixel(const rtc_camera* c, uint32_t x, uint32_t y, float origin[4], float direction[4]) {
    // camera.rs:60-74
    float x_offset = ((float)x + 0.5f) * c->pixel_size;
    float y_offset = ((float)y + 0.5f) * c->pixel_size;
    float p[4] = {c->half_width - x_offset, c->half_height - y_offset, -1.0f, 1.0f};
    float zero[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    float pixel[4];
    mat_vec4(c->inv, p, pixel);
    mat_vec4(c->inv, zero, origin);
    float diff[4] = {pixel[0] - origin[0], pixel[1] - origin[1], pixel[2] - origin[2], pixel[3] - origin[3]};
    norm4(diff, direction);
}

uint32_t rtc_partition_rows(uint32_t height, const rtc_partition* part) { return partition_rows(height, part); }

// ---- Canvas::to_ppm, canvas.rs:39-96 ------------------------------------------
static inline unsigned scale_color(float c) {
    // (c * 255).min(255).max(0) as u8: f32::min/max return the non-NaN operand; `as u8` truncates
    float v = fmaxf(fminf(c * 255.0f, 255.0f), 0.0f);
    return (unsigned)(uint8_t)v;
}

// One row of Canvas::to_ppm (canvas.rs:58-96) appended at `out` (WRITE) or merely measured; returns the end / the
// length.  The 70-column wrap starts afresh on every row (`current_line`, canvas.rs:69), so rows are independent.
extern "C++" {
template <bool WRITE>
static char* ppm_format_row(const float* row, uint32_t w, char* out) {
    size_t line_len = 0;
    auto put = [&](char ch) {
        if (WRITE) *out = ch;
        out++;
    };
    auto put_value = [&](unsigned v) {
        if (v >= 100) put((char)('0' + v / 100)), line_len++;
        if (v >= 10) put((char)('0' + (v / 10) % 10)), line_len++;
        put((char)('0' + v % 10));
        line_len++;
    };
    auto separator = [&]() {  // write_rgb_separator, canvas.rs:47-55 (70 - 3 = 67)
        if (line_len < 67) {
            put(' ');
            line_len++;
        } else {
            put('\n');
            line_len = 0;
        }
    };
    for (uint32_t col = 0; col < w; col++) {
        const float* p = row + (size_t)col * 3;
        put_value(scale_color(p[0]));
        separator();
        put_value(scale_color(p[1]));
        separator();
        put_value(scale_color(p[2]));
        if (col != w - 1) separator();
    }
    if (line_len != 0) put('\n');  // canvas.rs:90-93
    return out;
}
}  // extern "C++"

rtc_status rtc_to_ppm(const float* rgb, uint32_t w, uint32_t h, char** out_text, uint64_t* out_len) {
    if (!rgb || !out_text || !out_len) return fail(RTC_ERR_INVALID_ARG, "rtc_to_ppm: null argument");
    char head[64];
    const size_t head_len = (size_t)snprintf(head, sizeof(head), "P3\n%u %u\n255\n", w, h);
    // A 4096^2 frame is 175 MB of text -- 1.2 s on one core, a thousand renders.  A few threads measure their share of
    // the rows, then write it straight into its place in the one output buffer (fresh pages are the other cost, so
    // nothing is formatted into a scratch buffer and copied).
    unsigned n_threads = std::thread::hardware_concurrency();
    n_threads = n_threads == 0 ? 1 : (n_threads > 16 ? 16 : n_threads);
    if ((uint64_t)w * h < (1u << 16)) n_threads = 1;
    if (n_threads > h) n_threads = h ? h : 1;
    auto first_row = [&](unsigned t) { return (uint32_t)((uint64_t)h * t / n_threads); };
    auto on_all = [&](auto&& work) {
        std::vector<std::thread> workers;
        for (unsigned t = 1; t < n_threads; t++) workers.emplace_back(work, t);
        work(0u);
        for (auto& th : workers) th.join();
    };
    std::vector<size_t> len(n_threads, 0), at(n_threads, 0);
    on_all([&](unsigned t) {
        size_t n = 0;
        for (uint32_t row = first_row(t); row < first_row(t + 1); row++)
            n += (size_t)(ppm_format_row<false>(rgb + (size_t)row * w * 3, w, nullptr) - (char*)nullptr);
        len[t] = n;
    });
    size_t total = head_len;
    for (unsigned t = 0; t < n_threads; t++) at[t] = total, total += len[t];
    char* out = (char*)std::malloc(total + 1);
    if (!out) return fail(RTC_ERR_INVALID_ARG, "rtc_to_ppm: out of memory");
    std::memcpy(out, head, head_len);
    on_all([&](unsigned t) {
        char* p = out + at[t];
        for (uint32_t row = first_row(t); row < first_row(t + 1); row++) p = ppm_format_row<true>(rgb + (size_t)row * w * 3, w, p);
    });
    out[total] = 0;
    *out_text = out;
    *out_len = total;
    return RTC_OK;
}

void rtc_free(void* p) { std::free(p); }

}  // extern "C"
