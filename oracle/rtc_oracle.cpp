// rtc_oracle.cpp -- CPU ORACLE (test infrastructure, not product code).
//
// A literal, op-for-op f32 restatement of the render hot path of
// garfieldnate/ray_tracer_challenge.  It deliberately keeps the reference's
// *structure* (heap lists of intersections, a stable sort per ray, recursion,
// a linked set for refraction containers) so that it is an independent check
// of the HIP kernel, which is organised completely differently.
//
// Build: g++ -O2 -ffp-contract=off (see oracle/Makefile).  -ffp-contract=off
// matters: rustc never fuses a*b+c on x86-64, gcc would with -march=native.
// No -ffast-math.  f32 arithmetic is SSE single precision (no excess precision).
//
// Citations are file:line under /root/reference/lib/src.
#include "rtc_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------ tuple.rs
struct Tuple {
    float x, y, z, w;
};
inline Tuple point(float x, float y, float z) { return {x, y, z, 1.0f}; }   // tuple.rs:72-77
inline Tuple vector(float x, float y, float z) { return {x, y, z, 0.0f}; }  // tuple.rs:80-85
inline Tuple operator+(Tuple a, Tuple b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }  // :87-97
inline Tuple operator-(Tuple a, Tuple b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }  // :99-109
inline Tuple operator*(Tuple a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }          // :111-122
inline Tuple operator/(Tuple a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }          // :131-142
inline Tuple operator-(Tuple a) { return {-a.x, -a.y, -a.z, -a.w}; }                               // :144-155
// tuple.rs:29-33: powi(2) is x*x; the sum is left-associated and includes w.
inline float magnitude(Tuple a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w); }
// tuple.rs:34-43: divides (does not multiply by a reciprocal); w is kept.
inline Tuple norm(Tuple a) {
    float m = magnitude(a);
    return {a.x / m, a.y / m, a.z / m, a.w};
}
// tuple.rs:44-46
inline float dot(Tuple a, Tuple b) { return a.x * b.x + a.y * b.y + a.z * b.z + (a.w * b.w); }
// tuple.rs:47-55
inline Tuple cross(Tuple a, Tuple b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.0f};
}

// ------------------------------------------------------------------ color.rs
struct Color {
    float r, g, b;
};
inline Color operator+(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }  // color.rs:33-39
inline Color operator-(Color a, Color b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }  // color.rs:41-47
inline Color operator*(Color a, float s) { return {a.r * s, a.g * s, a.b * s}; }        // color.rs:50-56
inline Color operator*(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }  // color.rs:70-76
const Color BLACK = {0.0f, 0.0f, 0.0f};

// ----------------------------------------------------------------- matrix.rs
struct Matrix {
    int n;
    float d[4][4];
};
Matrix mat_new(int n) {  // matrix.rs:15-19
    Matrix m;
    m.n = n;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) m.d[r][c] = 0.0f;
    return m;
}
Matrix mat_from(const float* a, int n) {
    Matrix m = mat_new(n);
    for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++) m.d[r][c] = a[r * n + c];
    return m;
}
void mat_to(const Matrix& m, float* out) {
    for (int r = 0; r < m.n; r++)
        for (int c = 0; c < m.n; c++) out[r * m.n + c] = m.d[r][c];
}
Matrix mat4(float a, float b, float c, float d, float e, float f, float g, float h, float i, float j,
            float k, float l, float m_, float n_, float o, float p) {
    Matrix m = mat_new(4);
    float v[16] = {a, b, c, d, e, f, g, h, i, j, k, l, m_, n_, o, p};
    for (int r = 0; r < 4; r++)
        for (int cc = 0; cc < 4; cc++) m.d[r][cc] = v[r * 4 + cc];
    return m;
}
[[maybe_unused]] Matrix identity_4x4() { return mat4(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1); }  // matrix.rs:56-58
// matrix.rs:73-84
Tuple operator*(const Matrix& a, Tuple b) {
    float x = a.d[0][0] * b.x + a.d[0][1] * b.y + a.d[0][2] * b.z + a.d[0][3] * b.w;
    float y = a.d[1][0] * b.x + a.d[1][1] * b.y + a.d[1][2] * b.z + a.d[1][3] * b.w;
    float z = a.d[2][0] * b.x + a.d[2][1] * b.y + a.d[2][2] * b.z + a.d[2][3] * b.w;
    float w = a.d[3][0] * b.x + a.d[3][1] * b.y + a.d[3][2] * b.z + a.d[3][3] * b.w;
    return {x, y, z, w};
}
// matrix.rs:86-103
Matrix operator*(const Matrix& a, const Matrix& b) {
    Matrix m = mat_new(a.n);
    for (int r = 0; r < a.n; r++)
        for (int c = 0; c < a.n; c++)
            m.d[r][c] = a.d[r][0] * b.d[0][c] + a.d[r][1] * b.d[1][c] + a.d[r][2] * b.d[2][c] +
                        a.d[r][3] * b.d[3][c];
    return m;
}
Matrix transpose(const Matrix& a) {  // matrix.rs:134-143
    Matrix m = mat_new(a.n);
    for (int r = 0; r < a.n; r++)
        for (int c = 0; c < a.n; c++) m.d[c][r] = a.d[r][c];
    return m;
}
float cofactor(const Matrix& a, int row, int col);
// matrix.rs:145-160
float determinant(const Matrix& a) {
    if (a.n == 2) {
        return a.d[0][0] * a.d[1][1] - a.d[0][1] * a.d[1][0];
    }
    float det = 0.0f;
    for (int col = 0; col < a.n; col++) {
        float cf = cofactor(a, 0, col);
        det += cf * a.d[0][col];
    }
    return det;
}
// matrix.rs:163-182
Matrix submatrix(const Matrix& a, int remove_row, int remove_col) {
    Matrix m = mat_new(a.n - 1);
    int nr = 0;
    for (int r = 0; r < a.n; r++) {
        if (r == remove_row) continue;
        int nc = 0;
        for (int c = 0; c < a.n; c++) {
            if (c == remove_col) continue;
            m.d[nr][nc] = a.d[r][c];
            nc++;
        }
        nr++;
    }
    return m;
}
float minor_(const Matrix& a, int row, int col) { return determinant(submatrix(a, row, col)); }  // :194-196
float cofactor(const Matrix& a, int row, int col) {  // matrix.rs:184-192
    float m = minor_(a, row, col);
    return ((row + col) % 2 == 0) ? m : -m;
}
// matrix.rs:201-212
Matrix inverse(const Matrix& a) {
    float det = determinant(a);
    Matrix inv = mat_new(a.n);
    for (int row = 0; row < a.n; row++)
        for (int col = 0; col < a.n; col++) {
            float c = cofactor(a, row, col);
            inv.d[col][row] = c / det;
        }
    return inv;
}

// -------------------------------------------------------- transformations.rs
Matrix translation(float x, float y, float z) {  // :4-6
    return mat4(1, 0, 0, x, 0, 1, 0, y, 0, 0, 1, z, 0, 0, 0, 1);
}
Matrix scaling(float x, float y, float z) {  // :8-10
    return mat4(x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1);
}
// :12-43 -- f32::cos/f32::sin are libm cosf/sinf on Linux
Matrix rotation_x(float r) {
    float c = cosf(r), s = sinf(r);
    return mat4(1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1);
}
Matrix rotation_y(float r) {
    float c = cosf(r), s = sinf(r);
    return mat4(c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1);
}
Matrix rotation_z(float r) {
    float c = cosf(r), s = sinf(r);
    return mat4(c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1);
}
Matrix shearing(float xy, float xz, float yx, float yz, float zx, float zy) {  // :46-53
    return mat4(1, xy, xz, 0, yx, 1, yz, 0, zx, zy, 1, 0, 0, 0, 0, 1);
}
// :57-68
Matrix view_transform(Tuple from, Tuple to, Tuple approximate_up) {
    Tuple forward = norm(to - from);
    Tuple left = cross(forward, norm(approximate_up));
    Tuple true_up = cross(left, forward);
    Matrix orientation = mat4(left.x, left.y, left.z, 0, true_up.x, true_up.y, true_up.z, 0, -forward.x,
                              -forward.y, -forward.z, 0, 0, 0, 0, 1);
    return orientation * translation(-from.x, -from.y, -from.z);
}

// -------------------------------------------------------------------- ray.rs
struct Ray {
    Tuple origin, direction, direction_inverses;
};
Ray ray_new(Tuple o, Tuple d) {  // ray.rs:13-22 (reciprocals computed eagerly)
    return {o, d, vector(1.0f / d.x, 1.0f / d.y, 1.0f / d.z)};
}
Tuple position(const Ray& r, float t) { return r.origin + r.direction * t; }            // ray.rs:23-25
Ray transform(const Ray& r, const Matrix& m) { return ray_new(m * r.origin, m * r.direction); }  // :26-31
// ray.rs:42-44:  -(n*2*dot(in,n) - in)
Tuple reflect(Tuple in, Tuple n) { return -(n * 2.0f * dot(in, n) - in); }

// ----------------------------------------------------------- intersection.rs
struct Intersection {
    float distance;
    int object;
    float u, v;
    bool operator==(const Intersection& o) const {  // derived PartialEq, intersection.rs:4
        return distance == o.distance && object == o.object && u == o.u && v == o.v;
    }
};
// intersection.rs:30-35: filter(d >= 0).min_by(partial_cmp) -- min_by keeps the
// FIRST of equal minima (it only replaces on Ordering::Greater).
int hit_index(const std::vector<Intersection>& xs) {
    int best = -1;
    for (int i = 0; i < (int)xs.size(); i++) {
        if (!(xs[i].distance >= 0.0f)) continue;
        if (best < 0 || xs[best].distance > xs[i].distance) best = i;
    }
    return best;
}

// --------------------------------------------------------------- material.rs
// ------------------------------------------------------------- pattern/*.rs
// ---- pattern/uv.rs ----
struct UVPattern {
    int kind = 0;
    float width = 1, height = 1;
    Color colors[5];
    uint32_t iw = 0, ih = 0;
    std::vector<float> image;  // Canvas.data[y][x] as RGB f32
};
struct Pattern {
    int kind = RTCO_PATTERN_NONE;
    Color a{0, 0, 0}, b{0, 0, 0};
    Color distance{0, 0, 0};  // gradient.rs:17, sine_2d.rs:17: b - a, computed once in new()
    Matrix t_inverse;         // pattern.rs:52-54
    int uv_mapping = 0;
    std::vector<UVPattern> uv;
};
// Rust `f as i32` saturates (and maps NaN to 0); a bare C cast is undefined outside the i32 range.
inline int32_t rust_f32_as_i32(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
// Rust `f as usize`: saturating at 0 and usize::MAX, NaN -> 0
inline size_t rust_f32_as_usize(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 18446744073709551616.0f) return SIZE_MAX;
    return (size_t)f;
}
// f32::rem_euclid (core): r = self % rhs; if r < 0.0 { r + rhs.abs() } else { r }
inline float rem_euclid(float a, float rhs) {
    float r = fmodf(a, rhs);
    return r < 0.0f ? r + std::fabs(rhs) : r;
}
Color uv_color_at(const UVPattern& p, float u, float v) {
    switch (p.kind) {
        case RTCO_UV_CHECKERS: {  // uv.rs:45-55; i32 addition wraps in release, panics in debug: not reachable for u, v in [0, 1]
            int32_t u2 = rust_f32_as_i32(std::floor(u * p.width));
            int32_t v2 = rust_f32_as_i32(std::floor(v * p.height));
            return ((int32_t)((uint32_t)u2 + (uint32_t)v2)) % 2 == 0 ? p.colors[0] : p.colors[1];
        }
        case RTCO_UV_ALIGN_CHECK:  // uv.rs:145-165
            if (v > 0.8f) {
                if (u < 0.2f) return p.colors[1];
                if (u > 0.8f) return p.colors[2];
            } else if (v < 0.2f) {
                if (u < 0.2f) return p.colors[3];
                if (u > 0.8f) return p.colors[4];
            }
            return p.colors[0];
        case RTCO_UV_IMAGE: {  // uv.rs:366-376
            float vv = 1.0f - v;
            float x = u * (float)(p.iw - 1);
            float y = vv * (float)(p.ih - 1);
            size_t xi = rust_f32_as_usize(roundf(x)), yi = rust_f32_as_usize(roundf(y));
            if (xi >= p.iw || yi >= p.ih) return {0, 0, 0};  // the reference panics (index out of bounds)
            const float* px = &p.image[(yi * p.iw + xi) * 3];
            return {px[0], px[1], px[2]};
        }
    }
    return {0, 0, 0};
}
const float FRAC_1_PI = 0.318309886183790671537767526745028724f;  // std::f32::consts::FRAC_1_PI
const float PI_F = 3.14159265358979323846264338327950288f;
const float FRAC_1_2PI = 1.0f / (2.0f * PI_F);  // uv.rs:12
float calculate_u_from_azimuth(Tuple p) {  // uv.rs:107-113
    float theta = atan2f(p.x, p.z);
    float raw_u = theta * FRAC_1_2PI;
    return 1.0f - (raw_u + 0.5f);
}
void point_to_uv(int mapping, Tuple p, float* u, float* v) {
    switch (mapping) {
        case RTCO_MAP_SPHERICAL: {  // uv.rs:93-105
            *u = calculate_u_from_azimuth(p);
            float radius = magnitude(vector(p.x, p.y, p.z));
            float phi = acosf(p.y / radius);
            *v = 1.0f - phi * FRAC_1_PI;
            return;
        }
        case RTCO_MAP_PLANAR:  // uv.rs:182-186
            *u = rem_euclid(p.x, 1.0f);
            *v = rem_euclid(p.z, 1.0f);
            return;
        case RTCO_MAP_CYLINDRICAL:  // uv.rs:190-197
            *u = calculate_u_from_azimuth(p);
            *v = rem_euclid(p.y, 2.0f * PI_F) * FRAC_1_2PI;
            return;
    }
    *u = *v = 0.0f;
}
int face_from_point(Tuple p) {  // uv.rs:264-283: Front 0, Back 1, Left 2, Right 3, Up 4, Down 5
    float coord = fmaxf(fmaxf(std::fabs(p.x), std::fabs(p.y)), std::fabs(p.z));
    if (coord == p.x) return 3;
    if (coord == -p.x) return 2;
    if (coord == p.y) return 4;
    if (coord == -p.y) return 5;
    if (coord == p.z) return 0;
    return 1;
}
void cube_uv(int face, Tuple p, float* u, float* v) {  // uv.rs:285-319 (`%` on f32 is fmodf)
    switch (face) {
        case 0: *u = fmodf(p.x + 1.0f, 2.0f) / 2.0f; *v = fmodf(p.y + 1.0f, 2.0f) / 2.0f; return;  // front
        case 1: *u = fmodf(1.0f - p.x, 2.0f) / 2.0f; *v = fmodf(p.y + 1.0f, 2.0f) / 2.0f; return;  // back
        case 2: *u = fmodf(p.z + 1.0f, 2.0f) / 2.0f; *v = fmodf(p.y + 1.0f, 2.0f) / 2.0f; return;  // left
        case 3: *u = fmodf(1.0f - p.z, 2.0f) / 2.0f; *v = fmodf(p.y + 1.0f, 2.0f) / 2.0f; return;  // right
        case 4: *u = fmodf(p.x + 1.0f, 2.0f) / 2.0f; *v = fmodf(1.0f - p.z, 2.0f) / 2.0f; return;  // up
        default: *u = fmodf(p.x + 1.0f, 2.0f) / 2.0f; *v = fmodf(p.z + 1.0f, 2.0f) / 2.0f; return; // down
    }
}

Color pattern_color_at_world(const Pattern& p, Tuple pt) {
    switch (p.kind) {
        case RTCO_PATTERN_TEXTURE_MAP: {  // uv.rs:84-88
            float u, v;
            point_to_uv(p.uv_mapping, pt, &u, &v);
            return uv_color_at(p.uv[0], u, v);
        }
        case RTCO_PATTERN_CUBE_MAP: {  // uv.rs:248-261
            int face = face_from_point(pt);
            float u, v;
            cube_uv(face, pt, &u, &v);
            return uv_color_at(p.uv[face], u, v);
        }
        case RTCO_PATTERN_STRIPES:  // stripes.rs:39-45; Rust % keeps the dividend's sign, as C's does
            return rust_f32_as_i32(std::floor(pt.x)) % 2 == 0 ? p.a : p.b;
        case RTCO_PATTERN_GRADIENT: {  // gradient.rs:33-36
            float fraction = pt.x - std::floor(pt.x);
            return p.a + (p.distance * fraction);
        }
        case RTCO_PATTERN_RINGS:  // rings.rs:38-50
            return rust_f32_as_i32(std::floor(std::sqrt(pt.x * pt.x + pt.z * pt.z))) % 2 == 0 ? p.a : p.b;
        case RTCO_PATTERN_CHECKERS:  // checkers.rs:38-46
            return rust_f32_as_i32(std::floor(std::fabs(pt.x) + std::fabs(pt.y) + std::fabs(pt.z))) % 2 == 0 ? p.a : p.b;
        case RTCO_PATTERN_SINE2D: {  // sine_2d.rs:39-44; f32::cos is the platform cosf (glibc here)
            float cosine = cosf(pt.x + pt.z);
            float fraction = (-cosine + 1.0f) / 2.0f;
            return p.a + (p.distance * fraction);
        }
        case RTCO_PATTERN_TEST:  // pattern.rs:85-87
            return {pt.x, pt.y, pt.z};
    }
    return {0, 0, 0};
}
Pattern pattern_from(const rtco_pattern& c) {
    Pattern p;
    p.kind = c.kind;
    if (c.kind == RTCO_PATTERN_NONE) return p;
    p.a = {c.a[0], c.a[1], c.a[2]};
    p.b = {c.b[0], c.b[1], c.b[2]};
    p.distance = p.b - p.a;
    p.t_inverse = inverse(mat_from(c.transform, 4));
    p.uv_mapping = c.uv_mapping;
    if (c.kind == RTCO_PATTERN_TEXTURE_MAP || c.kind == RTCO_PATTERN_CUBE_MAP) {
        for (int k = 0; k < c.n_uv; k++) {
            const rtco_uv_pattern& src = c.uv[k];
            UVPattern u;
            u.kind = src.kind;
            u.width = src.width;
            u.height = src.height;
            for (int j = 0; j < 5; j++) u.colors[j] = {src.colors[j][0], src.colors[j][1], src.colors[j][2]};
            u.iw = src.image_width;
            u.ih = src.image_height;
            if (src.kind == RTCO_UV_IMAGE) u.image.assign(src.image_rgb, src.image_rgb + (size_t)u.iw * u.ih * 3);
            p.uv.push_back(std::move(u));
        }
    }
    return p;
}

struct Material {
    Color color;
    float ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    Pattern pattern;  // material.rs:50 Option<BoxedPattern>; kind NONE == None
};
Material material_from(const rtco_material& m) {
    Material mm;
    mm.color = {m.color[0], m.color[1], m.color[2]};
    mm.ambient = m.ambient;
    mm.diffuse = m.diffuse;
    mm.specular = m.specular;
    mm.shininess = m.shininess;
    mm.reflective = m.reflective;
    mm.transparency = m.transparency;
    mm.refractive_index = m.refractive_index;
    mm.pattern = pattern_from(m.pattern);
    return mm;
}

// ------------------------------------------------- shape/{shape,base_shape}.rs
const float CLOSE_TO_ZERO = 0.000001f;  // cylinder.rs:82
struct Shape {
    int kind;
    bool casts_shadow;
    bool closed;
    float min_y, max_y;
    Matrix t, t_inverse, t_inverse_transpose;  // base_shape.rs:56-60
    Material m;
    int id;
    Tuple p1, p2, p3, e1, e2, normal;  // triangle.rs:9-17
    Tuple n1, n2, n3;                  // smooth_triangle.rs:11-15
};
Shape shape_from(const rtco_shape& s, int id) {
    Shape sh;
    sh.kind = s.kind;
    sh.casts_shadow = s.casts_shadow != 0;
    sh.closed = s.closed != 0;
    sh.min_y = s.min_y;
    sh.max_y = s.max_y;
    sh.t = mat_from(s.transform, 4);
    sh.t_inverse = inverse(sh.t);                        // base_shape.rs:58
    sh.t_inverse_transpose = transpose(inverse(sh.t));   // base_shape.rs:59
    sh.m = material_from(s.material);
    sh.id = id;
    sh.p1 = sh.p2 = sh.p3 = sh.e1 = sh.e2 = sh.normal = sh.n1 = sh.n2 = sh.n3 = vector(0, 0, 0);
    if (s.kind == RTCO_TRIANGLE || s.kind == RTCO_SMOOTH_TRIANGLE) {  // triangle.rs:19-33
        sh.p1 = {s.p1[0], s.p1[1], s.p1[2], s.p1[3]};
        sh.p2 = {s.p2[0], s.p2[1], s.p2[2], s.p2[3]};
        sh.p3 = {s.p3[0], s.p3[1], s.p3[2], s.p3[3]};
        sh.e1 = sh.p2 - sh.p1;
        sh.e2 = sh.p3 - sh.p1;
        sh.normal = norm(cross(sh.e2, sh.e1));
        sh.n1 = {s.n1[0], s.n1[1], s.n1[2], s.n1[3]};
        sh.n2 = {s.n2[0], s.n2[1], s.n2[2], s.n2[3]};
        sh.n3 = {s.n3[0], s.n3[1], s.n3[2], s.n3[3]};
    }
    return sh;
}

// shape/cube.rs:90-129.  Rust f32::min/max ignore a NaN operand, like fminf/fmaxf.
bool aabb_intersection(const Ray& r, Tuple mn, Tuple mx, float* out_min, float* out_max) {
    float min_x = (mn.x - r.origin.x) * r.direction_inverses.x;
    float max_x = (mx.x - r.origin.x) * r.direction_inverses.x;
    float min_d = fminf(min_x, max_x);
    float max_d = fmaxf(min_x, max_x);
    float min_y = (mn.y - r.origin.y) * r.direction_inverses.y;
    float max_y = (mx.y - r.origin.y) * r.direction_inverses.y;
    min_d = fmaxf(min_d, fminf(min_y, max_y));
    max_d = fminf(max_d, fmaxf(min_y, max_y));
    float min_z = (mn.z - r.origin.z) * r.direction_inverses.z;
    float max_z = (mx.z - r.origin.z) * r.direction_inverses.z;
    min_d = fmaxf(min_d, fminf(min_z, max_z));
    max_d = fminf(max_d, fmaxf(min_z, max_z));
    if (max_d >= fmaxf(0.0f, min_d)) {
        *out_min = min_d;
        *out_max = max_d;
        return true;
    }
    return false;
}

// cylinder.rs:125-129
bool check_cap(const Ray& r, float t) {
    float x = r.origin.x + t * r.direction.x;
    float z = r.origin.z + t * r.direction.z;
    return (x * x + z * z) <= 1.0f + CLOSE_TO_ZERO;
}

// cone.rs:148-154
bool cone_check_cap(float radius, const Ray& r, float t) {
    float x = r.origin.x + t * r.direction.x;
    float z = r.origin.z + t * r.direction.z;
    return (x * x + z * z) <= radius + CLOSE_TO_ZERO;
}

void local_intersect(const Shape& s, const Ray& r, std::vector<Intersection>& out) {
    switch (s.kind) {
        case RTCO_SPHERE: {  // sphere.rs:47-70
            Tuple sphere_to_ray = r.origin - point(0, 0, 0);
            float a = dot(r.direction, r.direction);
            float b = 2.0f * dot(r.direction, sphere_to_ray);
            float c = dot(sphere_to_ray, sphere_to_ray) - 1.0f;
            float disc = b * b - 4.0f * a * c;
            if (disc < 0.0f) return;
            float two_a = 2.0f * a;
            float sq = std::sqrt(disc);
            out.push_back({(-b - sq) / two_a, s.id, 0.f, 0.f});
            out.push_back({(-b + sq) / two_a, s.id, 0.f, 0.f});
            return;
        }
        case RTCO_PLANE: {  // plane.rs:45-56 (f32::EPSILON * 10000.0)
            if (std::fabs(r.direction.y) < 1.1920929e-7f * 10000.0f) return;
            out.push_back({-r.origin.y / r.direction.y, s.id, 0.f, 0.f});
            return;
        }
        case RTCO_CUBE: {  // cube.rs:55-63
            float t0, t1;
            if (aabb_intersection(r, point(-1, -1, -1), point(1, 1, 1), &t0, &t1)) {
                out.push_back({t0, s.id, 0.f, 0.f});
                out.push_back({t1, s.id, 0.f, 0.f});
            }
            return;
        }
        case RTCO_CYLINDER: {  // cylinder.rs:52-59, 84-151
            size_t before = out.size();
            // intersect_sides :84-122
            do {
                float two_a = 2.0f * (r.direction.x * r.direction.x + r.direction.z * r.direction.z);
                if (std::fabs(two_a) < CLOSE_TO_ZERO) break;
                float b = 2.0f * (r.origin.x * r.direction.x + r.origin.z * r.direction.z);
                float c = r.origin.x * r.origin.x + r.origin.z * r.origin.z - 1.0f;
                float disc = b * b - 2.0f * two_a * c;
                if (disc < 0.0f) break;
                float sq = std::sqrt(disc);
                float d1 = (-b - sq) / two_a;
                float d2 = (-b + sq) / two_a;
                if (d1 > d2) std::swap(d1, d2);
                float y1 = r.origin.y + d1 * r.direction.y;
                if (s.min_y < y1 && y1 < s.max_y) out.push_back({d1, s.id, 0.f, 0.f});
                float y2 = r.origin.y + d2 * r.direction.y;
                if (s.min_y < y2 && y2 < s.max_y) out.push_back({d2, s.id, 0.f, 0.f});
            } while (false);
            // intersect_caps :132-151, only if the sides gave < 2 hits (:55-57)
            if (out.size() - before < 2 && s.closed) {
                float t = (s.min_y - r.origin.y) / r.direction.y;
                if (check_cap(r, t)) out.push_back({t, s.id, 0.f, 0.f});
                t = (s.max_y - r.origin.y) / r.direction.y;
                if (check_cap(r, t)) out.push_back({t, s.id, 0.f, 0.f});
            }
            return;
        }
        case RTCO_TRIANGLE:
        case RTCO_SMOOTH_TRIANGLE: {  // triangle.rs:45-68 (smooth_triangle.rs:37-39 delegates to it)
            Tuple dir_cross_e2 = cross(r.direction, s.e2);
            float determinant = dot(s.e1, dir_cross_e2);
            if (std::fabs(determinant) < 0.0000001f) return;
            float f = 1.0f / determinant;
            Tuple p1_to_origin = r.origin - s.p1;
            float u = f * dot(p1_to_origin, dir_cross_e2);
            if (u < 0.0f || u > 1.0f) return;
            Tuple origin_cross_e1 = cross(p1_to_origin, s.e1);
            float v = f * dot(r.direction, origin_cross_e1);
            if (v < 0.0f || (u + v) > 1.0f) return;
            float distance = f * dot(s.e2, origin_cross_e1);
            out.push_back({distance, s.id, u, v});
            return;
        }
        case RTCO_CONE: {  // cone.rs:52-57 (sides, then caps -- always, unlike the cylinder), :89-175
            do {  // intersect_sides :89-141
                float two_a = 2.0f * (r.direction.x * r.direction.x - r.direction.y * r.direction.y +
                                      r.direction.z * r.direction.z);
                float b = 2.0f * (r.origin.x * r.direction.x - r.origin.y * r.direction.y + r.origin.z * r.direction.z);
                // calc_c :143-146
                float c = r.origin.x * r.origin.x - r.origin.y * r.origin.y + r.origin.z * r.origin.z;
                if (std::fabs(two_a) < CLOSE_TO_ZERO) {
                    if (std::fabs(b) < CLOSE_TO_ZERO) break;
                    out.push_back({-c / (2.0f * b), s.id, 0.f, 0.f});  // no y-range check on this branch (:99-107)
                    break;
                }
                float disc = b * b - 2.0f * two_a * c;
                if (disc < 0.0f) break;
                float sq = std::sqrt(disc);
                float d1 = (-b - sq) / two_a;
                float d2 = (-b + sq) / two_a;
                if (d1 > d2) std::swap(d1, d2);
                float y1 = r.origin.y + d1 * r.direction.y;
                if (s.min_y < y1 && y1 < s.max_y) out.push_back({d1, s.id, 0.f, 0.f});
                float y2 = r.origin.y + d2 * r.direction.y;
                if (s.min_y < y2 && y2 < s.max_y) out.push_back({d2, s.id, 0.f, 0.f});
            } while (false);
            if (s.closed) {  // intersect_caps :156-175; check_cap :148-154 compares x^2+z^2 with |y| (not y^2)
                float t = (s.min_y - r.origin.y) / r.direction.y;
                if (cone_check_cap(std::fabs(s.min_y), r, t)) out.push_back({t, s.id, 0.f, 0.f});
                t = (s.max_y - r.origin.y) / r.direction.y;
                if (cone_check_cap(std::fabs(s.max_y), r, t)) out.push_back({t, s.id, 0.f, 0.f});
            }
            return;
        }
    }
}

Tuple local_norm_at(const Shape& s, Tuple p) {
    switch (s.kind) {
        case RTCO_SPHERE:  // sphere.rs:71-73
            return p - point(0, 0, 0);
        case RTCO_PLANE:  // plane.rs:57-59
            return vector(0, 1, 0);
        case RTCO_CUBE: {  // cube.rs:66-80
            float xa = std::fabs(p.x), ya = std::fabs(p.y), za = std::fabs(p.z);
            float max_c = fmaxf(xa, fmaxf(ya, za));
            if (xa == max_c) return vector(p.x, 0, 0);
            if (ya == max_c) return vector(0, p.y, 0);
            return vector(0, 0, p.z);
        }
        case RTCO_TEST_SHAPE:  // shape/test_shape.rs:38-45 (test double)
            return vector(2.0f * p.x, 3.0f * p.y, 4.0f * p.z);
        case RTCO_TRIANGLE:         // triangle.rs:70-72
        case RTCO_SMOOTH_TRIANGLE:  // the hit object of a smooth triangle is its inner Triangle (see rtc_oracle.h)
            return s.normal;
        case RTCO_CYLINDER: {  // cylinder.rs:62-72
            float dist_square = p.x * p.x + p.z * p.z;
            if (dist_square < 1.0f) {
                if (p.y >= s.max_y - CLOSE_TO_ZERO) return vector(0, 1, 0);
                if (p.y <= s.min_y + CLOSE_TO_ZERO) return vector(0, -1, 0);
            }
            return vector(p.x, 0, p.z);
        }
        case RTCO_CONE: {  // cone.rs:60-73 (the cap test uses radius 1, as written there)
            float dist_square = p.x * p.x + p.z * p.z;
            if (dist_square < 1.0f) {
                if (p.y >= s.max_y - CLOSE_TO_ZERO) return vector(0, 1, 0);
                if (p.y <= s.min_y + CLOSE_TO_ZERO) return vector(0, -1, 0);
            }
            float y = std::sqrt(p.x * p.x + p.z * p.z);
            if (p.y > 0.0f) y = -y;
            return vector(p.x, y, p.z);
        }
    }
    return vector(0, 0, 0);
}

// shape.rs:67-70
void shape_intersect(const Shape& s, const Ray& world_ray, std::vector<Intersection>& out, Ray* obj_ray = nullptr) {
    Ray object_ray = transform(world_ray, s.t_inverse);
    if (obj_ray) *obj_ray = object_ray;
    local_intersect(s, object_ray, out);
}
// shape.rs:72-154
Tuple normal_to_world(const Shape& s, Tuple object_normal) {
    Tuple world_normal = s.t_inverse_transpose * object_normal;
    world_normal.w = 0.0f;
    return norm(world_normal);
}
Tuple normal_at(const Shape& s, Tuple world_point) {
    Tuple object_point = s.t_inverse * world_point;
    Tuple object_normal = local_norm_at(s, object_point);
    return normal_to_world(s, object_normal);
}

// ------------------------------------------------------------------- jitter
// The reference draws from thread_rng() (rectangle_light.rs:46), which is
// irreproducible.  Pinned replacement (specified in DESIGN.md,
// implemented independently here and in the HIP kernel): a counter-based hash
// keyed by (pixel, path code, cell, draw) -- one 32-bit mix per draw -- mapped
// to the 23-bit grid of rand's OpenClosed01, ((h >> 9) + 1) * 2^-23 in (0, 1].
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
inline uint32_t jitter_hash(uint32_t seed, uint32_t pixel, uint32_t path, uint32_t cell, uint32_t draw) {
    uint32_t a = mix32(pixel ^ seed);
    uint32_t b = mix32(a + path * 0x9E3779B9u);
    return mix32(b + (2u * cell + draw) * 0x85EBCA6Bu);
}
inline float jitter_value(uint32_t h) { return (float)((h >> 9) + 1u) * 1.1920928955078125e-07f; }  // (0,1], 2^-23 steps

struct Light {
    int kind;
    Color intensity;
    Tuple position;  // point light position, or rectangle centre (rectangle_light.rs:57)
    Tuple corner, u_vec, v_vec;
    int u_steps, v_steps, cells;
    int jitter_mode;
    float jitter_const;
    uint32_t jitter_seed;
    std::vector<float> seq;
};

// ------------------------------------------------------------ bounding_box.rs
struct BBox {
    Tuple min, max;
};
BBox bbox_empty() {  // :13-20
    const float inf = std::numeric_limits<float>::infinity();
    return {point(inf, inf, inf), point(-inf, -inf, -inf)};
}
// :37-45.  Rust f32::min / max return the non-NaN operand, like fminf / fmaxf.
void bbox_add_point(BBox& b, Tuple p) {
    b.min.x = fminf(b.min.x, p.x);
    b.min.y = fminf(b.min.y, p.y);
    b.min.z = fminf(b.min.z, p.z);
    b.max.x = fmaxf(b.max.x, p.x);
    b.max.y = fmaxf(b.max.y, p.y);
    b.max.z = fmaxf(b.max.z, p.z);
}
void bbox_add(BBox& b, const BBox& o) {  // :47-50
    bbox_add_point(b, o.min);
    bbox_add_point(b, o.max);
}
bool between_inclusive(float v, float lo, float hi) { return v >= lo && v <= hi; }  // :23-31
bool bbox_contains_point(const BBox& b, Tuple p) {                                   // :52-56
    return between_inclusive(p.x, b.min.x, b.max.x) && between_inclusive(p.y, b.min.y, b.max.y) &&
           between_inclusive(p.z, b.min.z, b.max.z);
}
bool bbox_contains(const BBox& b, const BBox& o) {  // :58-60
    return bbox_contains_point(b, o.min) && bbox_contains_point(b, o.max);
}
BBox bbox_transform(const BBox& b, const Matrix& m) {  // :62-79 (eight corners, full 4x4 product each)
    BBox out = bbox_empty();
    const Tuple ps[8] = {b.min,
                         point(b.min.x, b.min.y, b.max.z),
                         point(b.min.x, b.max.y, b.min.z),
                         point(b.min.x, b.max.y, b.max.z),
                         point(b.max.x, b.min.y, b.min.z),
                         point(b.max.x, b.min.y, b.max.z),
                         point(b.max.x, b.max.y, b.min.z),
                         b.max};
    for (const Tuple& p : ps) bbox_add_point(out, m * p);
    return out;
}
bool bbox_intersects(const BBox& b, const Ray& r) {  // :81-83
    float t0, t1;
    return aabb_intersection(r, b.min, b.max, &t0, &t1);
}
void bbox_split(const BBox& b, BBox* left, BBox* right) {  // :85-113
    float dx = b.max.x - b.min.x;
    float dy = b.max.y - b.min.y;
    float dz = b.max.z - b.min.z;
    float greatest = fmaxf(fmaxf(dx, dy), dz);
    float x0 = b.min.x, y0 = b.min.y, z0 = b.min.z;
    float x1 = b.max.x, y1 = b.max.y, z1 = b.max.z;
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
    *left = {b.min, point(x1, y1, z1)};
    *right = {point(x0, y0, z0), b.max};
}
// Shape::bounding_box per kind: sphere.rs:75-80, plane.rs:61-66, cube.rs:82-87, cylinder.rs:74-79,
// cone.rs:75-85, test_shape.rs:47-53
BBox shape_bounding_box(const Shape& s) {
    const float inf = std::numeric_limits<float>::infinity();
    switch (s.kind) {
        case RTCO_PLANE:
            return {point(-inf, 0, -inf), point(inf, 0, inf)};
        case RTCO_CYLINDER:
            return {point(-1, s.min_y, -1), point(1, s.max_y, 1)};
        case RTCO_CONE: {
            float limit = fmaxf(std::fabs(s.min_y), std::fabs(s.max_y));
            return {point(-limit, s.min_y, -limit), point(limit, s.max_y, limit)};
        }
        case RTCO_TRIANGLE:
        case RTCO_SMOOTH_TRIANGLE: {  // triangle.rs:74-80
            BBox b = bbox_empty();
            bbox_add_point(b, s.p1);
            bbox_add_point(b, s.p2);
            bbox_add_point(b, s.p3);
            return b;
        }
        default:
            return {point(-1, -1, -1), point(1, 1, 1)};
    }
}

// ------------------------------------------------------------ shape/group.rs
// Nodes of shape trees live in one arena; a node is a leaf (index into `leaves`) or a GroupShape.
struct TNode {
    bool group = false;
    int leaf = -1;
    Matrix t, t_inverse;  // the group's own BaseShape fields (base_shape.rs:16-17); leaves keep theirs in Shape
    std::vector<int> children;
    bool has_cache = false;  // cached_bounding_box: RefCell<Option<BoundingBox>>, group.rs:15 -- never invalidated
    BBox cache;
};
struct Tree {
    std::vector<Shape> leaves;
    std::vector<TNode> nodes;
};
void leaf_set_transformation(Shape& sh, const Matrix& t) {  // base_shape.rs:56-60
    sh.t = t;
    sh.t_inverse = inverse(t);
    sh.t_inverse_transpose = transpose(inverse(t));
}
const Matrix& node_transformation(const Tree& tr, int n) {
    return tr.nodes[n].group ? tr.nodes[n].t : tr.leaves[tr.nodes[n].leaf].t;
}
void node_set_transformation(Tree& tr, int n, const Matrix& t) {
    TNode& nd = tr.nodes[n];
    if (!nd.group) {
        leaf_set_transformation(tr.leaves[nd.leaf], t);
        return;
    }
    // group.rs:101-114: re-bake the children, then store the group's own transform
    if (!nd.children.empty()) {
        Matrix child_transformer = t * nd.t_inverse;
        std::vector<int> kids = nd.children;
        for (int c : kids) {
            Matrix old_child_transform = node_transformation(tr, c);
            node_set_transformation(tr, c, child_transformer * old_child_transform);
        }
    }
    TNode& nd2 = tr.nodes[n];
    nd2.t = t;
    nd2.t_inverse = inverse(t);
}
void group_add_child(Tree& tr, int g, int child) {  // group.rs:39-44
    Matrix old_child_transform = node_transformation(tr, child);
    node_set_transformation(tr, child, tr.nodes[g].t * old_child_transform);
    tr.nodes[g].children.push_back(child);
}
void node_set_material(Tree& tr, int n, const Material& m) {  // group.rs:96-100; base_shape.rs:64-66
    if (!tr.nodes[n].group) {
        tr.leaves[tr.nodes[n].leaf].m = m;
        return;
    }
    for (int c : tr.nodes[n].children) node_set_material(tr, c, m);
}
BBox node_parent_space_bounding_box(Tree& tr, int n);
BBox node_bounding_box(Tree& tr, int n) {
    TNode& nd = tr.nodes[n];
    if (!nd.group) return shape_bounding_box(tr.leaves[nd.leaf]);
    if (!nd.has_cache) {  // group.rs:138-151
        BBox b = bbox_empty();
        std::vector<int> kids = nd.children;
        for (int c : kids) bbox_add(b, node_parent_space_bounding_box(tr, c));
        tr.nodes[n].cache = b;
        tr.nodes[n].has_cache = true;
    }
    return tr.nodes[n].cache;
}
BBox node_parent_space_bounding_box(Tree& tr, int n) {
    if (tr.nodes[n].group) return node_bounding_box(tr, n);  // group.rs:153-155
    const Shape& sh = tr.leaves[tr.nodes[n].leaf];
    return bbox_transform(shape_bounding_box(sh), sh.t);     // shape.rs:162-164
}
int tree_new_group(Tree& tr) {
    TNode nd;
    nd.group = true;
    nd.t = identity_4x4();
    nd.t_inverse = identity_4x4();  // BaseShape::default(): no inversion
    tr.nodes.push_back(nd);
    return (int)tr.nodes.size() - 1;
}
void group_partition_children(Tree& tr, int g, std::vector<int>* left, std::vector<int>* right) {  // group.rs:46-64
    BBox lb, rb;
    bbox_split(node_bounding_box(tr, g), &lb, &rb);
    std::vector<int> kids;
    kids.swap(tr.nodes[g].children);
    std::vector<int> keep;
    for (int c : kids) {
        BBox cb = node_parent_space_bounding_box(tr, c);
        if (bbox_contains(lb, cb)) left->push_back(c);
        else if (bbox_contains(rb, cb)) right->push_back(c);
        else keep.push_back(c);
    }
    tr.nodes[g].children = keep;
}
void group_make_subgroup(Tree& tr, int g, const std::vector<int>& kids) {  // group.rs:66-73
    if (kids.size() == 1) {
        tr.nodes[g].children.push_back(kids[0]);
    } else {
        int sub = tree_new_group(tr);  // GroupShape::with_children: identity transform, nothing re-baked
        tr.nodes[sub].children = kids;
        tr.nodes[g].children.push_back(sub);
    }
}
void node_divide(Tree& tr, int n, size_t threshold) {  // group.rs:157-172; shape.rs:167 (no-op for leaves)
    if (!tr.nodes[n].group) return;
    if (threshold <= tr.nodes[n].children.size()) {
        std::vector<int> left, right;
        group_partition_children(tr, n, &left, &right);
        if (!left.empty()) group_make_subgroup(tr, n, left);
        if (!right.empty()) group_make_subgroup(tr, n, right);
    }
    std::vector<int> kids = tr.nodes[n].children;
    for (int c : kids) node_divide(tr, c, threshold);
}
// Shape::intersect for a node: group.rs:115-133 (the ray is NOT transformed: transforms are baked into
// the children) / shape.rs:67-70
void node_intersect(Tree& tr, int n, const Ray& r, std::vector<Intersection>& out) {
    if (!tr.nodes[n].group) {
        shape_intersect(tr.leaves[tr.nodes[n].leaf], r, out, nullptr);
        return;
    }
    BBox b = node_bounding_box(tr, n);
    if (!bbox_intersects(b, r)) return;
    std::vector<int> kids = tr.nodes[n].children;
    for (int c : kids) node_intersect(tr, c, r, out);
}

}  // namespace

struct rtco_world {
    std::vector<Shape> objects;  // every leaf shape; Intersection.object indexes this
    Tree tree;                   // used when `roots` is non-empty: World.objects = the root nodes (leaves or groups)
    std::vector<int> roots;
    Light light;
    // jitter state
    size_t seq_cursor = 0;
    uint32_t pixel = 0;
    uint64_t rays = 0;
};

namespace {

using World = rtco_world;

float jitter(World& w, uint32_t path, uint32_t cell, uint32_t draw) {
    switch (w.light.jitter_mode) {
        case RTCO_JITTER_CONSTANT:
            return w.light.jitter_const;  // test/utils.rs:15-17
        case RTCO_JITTER_CYCLE: {         // test/utils.rs:19-24
            float v = w.light.seq[w.seq_cursor % w.light.seq.size()];
            w.seq_cursor++;
            return v;
        }
        default:
            return jitter_value(jitter_hash(w.light.jitter_seed, w.pixel, path, cell, draw));
    }
}

// rectangle_light.rs:60-66
Tuple point_on_light(World& w, int u, int v, uint32_t path) {
    const Light& l = w.light;
    uint32_t cell = (uint32_t)(v * l.u_steps + u);
    float jitter1 = jitter(w, path, cell, 0);
    float jitter2 = jitter(w, path, cell, 1);
    return l.corner + l.u_vec * ((float)u + jitter1) + l.v_vec * ((float)v + jitter2);
}

// world.rs:52-60
std::vector<Intersection> intersect(World& w, const Ray& r) {
    w.rays++;
    std::vector<Intersection> xs;
    if (w.roots.empty()) {
        for (const Shape& o : w.objects) shape_intersect(o, r, xs);
    } else {
        for (int n : w.roots) node_intersect(w.tree, n, r, xs);
    }
    std::stable_sort(xs.begin(), xs.end(),
                     [](const Intersection& a, const Intersection& b) { return a.distance < b.distance; });
    return xs;
}

// world.rs:104-119
bool is_shadowed(World& w, Tuple light_position, Tuple p) {
    Tuple light_to_point = light_position - p;
    float distance = magnitude(light_to_point);
    Tuple direction = norm(light_to_point);
    Ray r = ray_new(p, direction);
    std::vector<Intersection> xs = intersect(w, r);
    int h = hit_index(xs);
    if (h < 0) return false;
    return w.objects[xs[h].object].casts_shadow && xs[h].distance < distance;
}

// point_light.rs:28-34, rectangle_light.rs:76-88
float intensity_at(World& w, Tuple p, uint32_t path) {
    const Light& l = w.light;
    if (l.kind == RTCO_LIGHT_POINT) {
        return is_shadowed(w, l.position, p) ? 0.0f : 1.0f;
    }
    float total = 0.0f;
    for (int v = 0; v < l.v_steps; v++)
        for (int u = 0; u < l.u_steps; u++) {
            Tuple lp = point_on_light(w, u, v, path);
            if (!is_shadowed(w, lp, p)) total += 1.0f;
        }
    return total / (float)l.cells;
}

// light/phong_lighting.rs:12-63 (pattern branch :24-27 out of scope)
// pattern.rs:15-19
Color pattern_color_at_object(const Pattern& pat, const Shape& object, Tuple world_point) {
    Tuple object_point = object.t_inverse * world_point;
    Tuple pattern_point = pat.t_inverse * object_point;
    return pattern_color_at_world(pat, pattern_point);
}

Color phong_lighting(const Shape& object, const Material& m, const Light& light, Tuple p, Tuple eye, Tuple n,
                     float light_intensity) {
    // phong_lighting.rs:24-28
    Color material_color = m.pattern.kind != RTCO_PATTERN_NONE ? pattern_color_at_object(m.pattern, object, p) : m.color;
    Color effective = material_color * light.intensity;
    Color ambient = effective * m.ambient;
    if (light_intensity == 0.0f) return ambient;
    Tuple to_light = norm(light.position - p);
    float light_normal_cosine = dot(to_light, n);
    Color diffuse, specular;
    if (light_normal_cosine < 0.0f) {
        diffuse = BLACK;
        specular = BLACK;
    } else {
        diffuse = effective * m.diffuse * light_normal_cosine;
        Tuple surface_reflection = reflect(-to_light, n);
        float reflection_eye_cosine = dot(surface_reflection, eye);
        if (reflection_eye_cosine <= 0.0f) {
            specular = BLACK;
        } else {
            float factor = powf(reflection_eye_cosine, m.shininess);  // f32::powf -> libm powf
            specular = light.intensity * m.specular * factor;
        }
    }
    return ambient + (diffuse + specular) * light_intensity;
}

struct Comps {  // world.rs:165-182
    float distance;
    int object;
    Tuple point, eye, reflectv, normal;
    bool inside;
    Tuple over_point, under_point;
    float n1, n2;
};

const float SELF_EPS = 1.1920929e-7f * 10000.0f;  // world.rs:210

// world.rs:212-283
Comps precompute_values(World& w, const Ray& r, const Intersection& hit, const std::vector<Intersection>& xs) {
    Comps c;
    c.point = position(r, hit.distance);
    Tuple n = normal_at(w.objects[hit.object], c.point);
    c.eye = -r.direction;
    c.reflectv = reflect(r.direction, n);
    if (dot(n, c.eye) < 0.0f) {
        c.inside = true;
        n = -n;
    } else {
        c.inside = false;
    }
    c.over_point = c.point + n * SELF_EPS;
    c.under_point = c.point - n * SELF_EPS;
    float n1 = NAN, n2 = NAN;
    // LinkedHashSet semantics (world.rs:239): insertion-ordered; remove -> bool; back() = newest.
    std::vector<int> containing;
    const float default_index = 1.0f;  // REFRACTION_VACCUM, constants.rs:6
    for (const Intersection& i : xs) {
        if (i == hit) {
            n1 = containing.empty() ? default_index : w.objects[containing.back()].m.refractive_index;
        }
        auto it = std::find(containing.begin(), containing.end(), i.object);
        if (it != containing.end())
            containing.erase(it);
        else
            containing.push_back(i.object);
        if (i == hit) {
            n2 = containing.empty() ? default_index : w.objects[containing.back()].m.refractive_index;
            break;
        }
    }
    c.distance = hit.distance;
    c.object = hit.object;
    c.normal = n;
    c.n1 = n1;
    c.n2 = n2;
    return c;
}

// world.rs:285-303.  powi(5): x2=x*x; x4=x2*x2; x*x4 (LLVM's expansion and
// compiler-rt's __powisf2 agree on this product).
float schlick_reflectance(const Comps& c) {
    float cosine = dot(c.eye, c.normal);
    if (c.n1 > c.n2) {
        float n = c.n1 / c.n2;
        float sin2_refracted = n * n * (1.0f - cosine * cosine);
        if (sin2_refracted > 1.0f) return 1.0f;
        cosine = std::sqrt(1.0f - sin2_refracted);
    }
    float q = (c.n1 - c.n2) / (c.n1 + c.n2);
    float r0 = q * q;
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x4 = x2 * x2;
    float x5 = x * x4;
    return r0 + (1.0f - r0) * x5;
}

Color color_at(World& w, const Ray& r, int remaining, uint32_t path);

// world.rs:121-133
Color reflected_color(World& w, const Comps& c, int remaining, uint32_t path) {
    const Material& m = w.objects[c.object].m;
    if (m.reflective == 0.0f || remaining < 1) return BLACK;
    Ray rr = ray_new(c.over_point, c.reflectv);
    Color col = color_at(w, rr, remaining - 1, path * 2u);
    return col * m.reflective;
}

// world.rs:135-162, 196-207
Color refracted_color(World& w, const Comps& c, int remaining, uint32_t path) {
    const Material& m = w.objects[c.object].m;
    if (m.transparency == 0.0f || remaining == 0) return BLACK;
    float n_ratio = c.n1 / c.n2;
    float cos_incoming = dot(c.eye, c.normal);
    float sin2 = n_ratio * n_ratio * (1.0f - cos_incoming * cos_incoming);
    if (sin2 > 1.0f) return BLACK;
    float cos_refracted = std::sqrt(1.0f - sin2);
    Tuple dir = c.normal * (n_ratio * cos_incoming - cos_refracted) - (c.eye * n_ratio);
    Ray rr = ray_new(c.under_point, dir);
    return color_at(w, rr, remaining - 1, path * 2u + 1u) * m.transparency;
}

// world.rs:62-86
Color shade_hit(World& w, const Comps& c, int remaining, uint32_t path) {
    const Material& m = w.objects[c.object].m;
    Color surface = phong_lighting(w.objects[c.object], m, w.light, c.over_point, c.eye, c.normal, intensity_at(w, c.over_point, path));
    Color reflected = reflected_color(w, c, remaining, path);
    Color refracted = refracted_color(w, c, remaining, path);
    if (m.reflective > 0.0f && m.transparency > 0.0f) {
        float reflectance = schlick_reflectance(c);
        return surface + reflected * reflectance + refracted * (1.0f - reflectance);
    }
    return surface + reflected + refracted;
}

// world.rs:88-101
Color color_at(World& w, const Ray& r, int remaining, uint32_t path) {
    std::vector<Intersection> xs = intersect(w, r);
    if (xs.empty()) return BLACK;
    int h = hit_index(xs);
    if (h < 0) return BLACK;
    Comps c = precompute_values(w, r, xs[h], xs);
    return shade_hit(w, c, remaining, path);
}

struct Camera {  // camera.rs:8-21
    uint32_t w, h;
    float fov, half_w, half_h, pixel_size;
    Matrix inv;
};
// camera.rs:23-56 (f32::tan -> libm tanf)
Camera camera_new(uint32_t w, uint32_t h, float fov, const Matrix& transform) {
    float half_view = tanf(fov / 2.0f);
    float aspect = (float)w / (float)h;
    Camera c;
    if (aspect >= 1.0f) {
        c.half_w = half_view;
        c.half_h = half_view / aspect;
    } else {
        c.half_w = half_view * aspect;
        c.half_h = half_view;
    }
    c.pixel_size = (c.half_w * 2.0f) / (float)w;
    c.w = w;
    c.h = h;
    c.fov = fov;
    c.inv = inverse(transform);
    return c;
}
// camera.rs:60-74
Ray ray_for_pixel(const Camera& c, uint32_t x, uint32_t y) {
    float x_offset = ((float)x + 0.5f) * c.pixel_size;
    float y_offset = ((float)y + 0.5f) * c.pixel_size;
    float world_x = c.half_w - x_offset;
    float world_y = c.half_h - y_offset;
    Tuple pixel = c.inv * point(world_x, world_y, -1);
    Tuple origin = c.inv * point(0, 0, 0);
    Tuple direction = norm(pixel - origin);
    return ray_new(origin, direction);
}
Camera camera_from(const rtco_camera* c) {
    Camera k;
    k.w = c->width;
    k.h = c->height;
    k.fov = c->field_of_view;
    k.half_w = c->half_width;
    k.half_h = c->half_height;
    k.pixel_size = c->pixel_size;
    k.inv = mat_from(c->transform_inverse, 4);
    return k;
}

inline Tuple T(const float* v) { return {v[0], v[1], v[2], v[3]}; }
inline void put(Tuple t, float* o) {
    o[0] = t.x;
    o[1] = t.y;
    o[2] = t.z;
    o[3] = t.w;
}
inline void putc(Color c, float* o) {
    o[0] = c.r;
    o[1] = c.g;
    o[2] = c.b;
}
Comps comps_from(const rtco_comps* c) {
    Comps k;
    k.distance = c->distance;
    k.object = c->object;
    k.point = T(c->point);
    k.eye = T(c->eye);
    k.reflectv = T(c->reflectv);
    k.normal = T(c->normal);
    k.over_point = T(c->over_point);
    k.under_point = T(c->under_point);
    k.inside = c->inside != 0;
    k.n1 = c->n1;
    k.n2 = c->n2;
    return k;
}

// canvas.rs:39-43: (c*255).min(255).max(0) as u8 -- f32::min/max drop NaN, `as u8` truncates.
uint8_t scale_color(float c) {
    float v = fmaxf(fminf(c * 255.0f, 255.0f), 0.0f);
    return (uint8_t)v;
}

// camera.rs:76-91 restricted to rows [y0,y1); the loop bounds keep the
// reference's off-by-one (last row and last column are never traced).
uint64_t render_rows_serial(World& w, const Camera& c, int depth, uint32_t y0, uint32_t y1, float* out) {
    uint64_t before = w.rays;
    uint32_t yend = std::min(y1, c.h - 1);
    for (uint32_t y = y0; y < yend; y++)
        for (uint32_t x = 0; x < c.w - 1; x++) {
            Ray r = ray_for_pixel(c, x, y);
            w.pixel = y * c.w + x;
            Color col = color_at(w, r, depth, 1u);
            putc(col, out + ((size_t)y * c.w + x) * 3);
        }
    return w.rays - before;
}

}  // namespace

extern "C" {
static void fill_light(rtco_world* w, const rtco_light* light);

float rtco_magnitude(const float v[4]) { return magnitude(T(v)); }
void rtco_norm(const float v[4], float out[4]) { put(norm(T(v)), out); }
float rtco_dot(const float a[4], const float b[4]) { return dot(T(a), T(b)); }
void rtco_cross(const float a[4], const float b[4], float out[4]) { put(cross(T(a), T(b)), out); }
void rtco_reflect(const float in[4], const float n[4], float out[4]) { put(reflect(T(in), T(n)), out); }
void rtco_position(const float o[4], const float d[4], float t, float out[4]) {
    put(position(ray_new(T(o), T(d)), t), out);
}

void rtco_mat_mul(const float a[16], const float b[16], float out[16]) {
    mat_to(mat_from(a, 4) * mat_from(b, 4), out);
}
void rtco_mat_vec(const float a[16], const float v[4], float out[4]) { put(mat_from(a, 4) * T(v), out); }
void rtco_mat_transpose(const float* a, int n, float* out) { mat_to(transpose(mat_from(a, n)), out); }
float rtco_mat_determinant(const float* a, int n) { return determinant(mat_from(a, n)); }
void rtco_mat_submatrix(const float* a, int n, int row, int col, float* out) {
    mat_to(submatrix(mat_from(a, n), row, col), out);
}
float rtco_mat_minor(const float* a, int n, int row, int col) { return minor_(mat_from(a, n), row, col); }
float rtco_mat_cofactor(const float* a, int n, int row, int col) { return cofactor(mat_from(a, n), row, col); }
void rtco_mat_inverse(const float* a, int n, float* out) { mat_to(inverse(mat_from(a, n)), out); }

void rtco_translation(float x, float y, float z, float out[16]) { mat_to(translation(x, y, z), out); }
void rtco_scaling(float x, float y, float z, float out[16]) { mat_to(scaling(x, y, z), out); }
void rtco_rotation_x(float r, float out[16]) { mat_to(rotation_x(r), out); }
void rtco_rotation_y(float r, float out[16]) { mat_to(rotation_y(r), out); }
void rtco_rotation_z(float r, float out[16]) { mat_to(rotation_z(r), out); }
void rtco_shearing(float xy, float xz, float yx, float yz, float zx, float zy, float out[16]) {
    mat_to(shearing(xy, xz, yx, yz, zx, zy), out);
}
void rtco_view_transform(const float from[4], const float to[4], const float up[4], float out[16]) {
    mat_to(view_transform(T(from), T(to), T(up)), out);
}

void rtco_camera_new(uint32_t w, uint32_t h, float fov, const float transform[16], rtco_camera* out) {
    Camera c = camera_new(w, h, fov, mat_from(transform, 4));
    out->width = w;
    out->height = h;
    out->field_of_view = fov;
    out->half_width = c.half_w;
    out->half_height = c.half_h;
    out->pixel_size = c.pixel_size;
    mat_to(c.inv, out->transform_inverse);
}
void rtco_ray_for_pixel(const rtco_camera* c, uint32_t x, uint32_t y, float o[4], float d[4]) {
    Ray r = ray_for_pixel(camera_from(c), x, y);
    put(r.origin, o);
    put(r.direction, d);
}

int rtco_local_intersect(const rtco_shape* s, const float o[4], const float d[4], float ts[4]) {
    Shape sh = shape_from(*s, 0);
    std::vector<Intersection> xs;
    local_intersect(sh, ray_new(T(o), T(d)), xs);
    for (size_t i = 0; i < xs.size() && i < 4; i++) ts[i] = xs[i].distance;
    return (int)xs.size();
}
void rtco_local_normal_at(const rtco_shape* s, const float p[4], float out[4]) {
    put(local_norm_at(shape_from(*s, 0), T(p)), out);
}
int rtco_local_intersect_uv(const rtco_shape* s, const float o[4], const float d[4], float ts[4], float us[4],
                            float vs[4]) {
    Shape sh = shape_from(*s, 0);
    std::vector<Intersection> xs;
    local_intersect(sh, ray_new(T(o), T(d)), xs);
    for (size_t i = 0; i < xs.size() && i < 4; i++) {
        ts[i] = xs[i].distance;
        us[i] = xs[i].u;
        vs[i] = xs[i].v;
    }
    return (int)xs.size();
}
void rtco_normal_at_uv(const rtco_shape* s, const float world_point[4], float u, float v, float out[4]) {
    Shape sh = shape_from(*s, 0);
    if (sh.kind != RTCO_SMOOTH_TRIANGLE) {
        put(normal_at(sh, T(world_point)), out);
        return;
    }
    // smooth_triangle.rs:41-43, then shape.rs:72-154 normal_to_world
    Tuple object_normal = sh.n2 * u + sh.n3 * v + sh.n1 * (1.0f - u - v);
    put(normal_to_world(sh, object_normal), out);
}
void rtco_triangle_fields(const rtco_shape* s, float e1[4], float e2[4], float normal[4]) {
    Shape sh = shape_from(*s, 0);
    put(sh.e1, e1);
    put(sh.e2, e2);
    put(sh.normal, normal);
}
int rtco_shape_intersect(const rtco_shape* s, const float o[4], const float d[4], float ts[4], float obj_o[4],
                         float obj_d[4]) {
    Shape sh = shape_from(*s, 0);
    std::vector<Intersection> xs;
    Ray orr;
    shape_intersect(sh, ray_new(T(o), T(d)), xs, &orr);
    put(orr.origin, obj_o);
    put(orr.direction, obj_d);
    for (size_t i = 0; i < xs.size() && i < 4; i++) ts[i] = xs[i].distance;
    return (int)xs.size();
}
void rtco_normal_at(const rtco_shape* s, const float world_point[4], float out[4]) {
    put(normal_at(shape_from(*s, 0), T(world_point)), out);
}
int rtco_aabb_intersection(const float o[4], const float d[4], const float mn[4], const float mx[4],
                           float out_t[2]) {
    return aabb_intersection(ray_new(T(o), T(d)), T(mn), T(mx), &out_t[0], &out_t[1]) ? 1 : 0;
}
int rtco_hit(const float* ts, int n) {
    std::vector<Intersection> xs;
    for (int i = 0; i < n; i++) xs.push_back({ts[i], 0, 0.f, 0.f});
    return hit_index(xs);
}

rtco_world* rtco_world_new(const rtco_shape* shapes, int n, const rtco_light* light) {
    rtco_world* w = new rtco_world();
    for (int i = 0; i < n; i++) w->objects.push_back(shape_from(shapes[i], i));
    fill_light(w, light);
    return w;
}
static void fill_light(rtco_world* w, const rtco_light* light) {
    Light& l = w->light;
    l.kind = light->kind;
    l.intensity = {light->intensity[0], light->intensity[1], light->intensity[2]};
    l.jitter_mode = light->jitter_mode;
    l.jitter_const = light->jitter_const;
    l.jitter_seed = light->jitter_seed;
    l.u_steps = l.v_steps = l.cells = 1;
    l.corner = l.u_vec = l.v_vec = vector(0, 0, 0);
    if (l.kind == RTCO_LIGHT_POINT) {
        l.position = T(light->position);
    } else {
        // rectangle_light.rs:33-58
        Tuple u = T(light->u_vec), v = T(light->v_vec);
        l.corner = T(light->corner);
        l.u_steps = light->u_steps;
        l.v_steps = light->v_steps;
        l.u_vec = u / (float)l.u_steps;
        l.v_vec = v / (float)l.v_steps;
        l.cells = l.u_steps * l.v_steps;
        l.position = l.corner + (u / 2.0f) + (v / 2.0f);
        if (light->jitter_seq && light->jitter_seq_len > 0)
            l.seq.assign(light->jitter_seq, light->jitter_seq + light->jitter_seq_len);
    }
}
// ---- shape trees (shape/group.rs).  One process-wide arena: this is test infrastructure. ----
static Tree g_arena;

int rtco_node_shape(const rtco_shape* s) {  // Shape::new() + set_transformation + set_material + pub fields
    Shape sh = shape_from(*s, (int)g_arena.leaves.size());
    g_arena.leaves.push_back(sh);
    TNode nd;
    nd.leaf = sh.id;
    g_arena.nodes.push_back(nd);
    return (int)g_arena.nodes.size() - 1;
}
int rtco_node_group(void) { return tree_new_group(g_arena); }                        // GroupShape::new()
int rtco_node_group_with_children(const int* children, int n) {                      // group.rs:23-27
    int g = tree_new_group(g_arena);
    g_arena.nodes[g].children.assign(children, children + n);
    return g;
}
void rtco_node_add_child(int group, int child) { group_add_child(g_arena, group, child); }
void rtco_node_set_transformation(int node, const float t[16]) { node_set_transformation(g_arena, node, mat_from(t, 4)); }
void rtco_node_set_material(int node, const rtco_material* m) { node_set_material(g_arena, node, material_from(*m)); }
void rtco_node_divide(int node, uint32_t threshold) { node_divide(g_arena, node, threshold); }
int rtco_node_is_group(int node) { return g_arena.nodes[node].group ? 1 : 0; }
int rtco_node_children(int node, int* out, int cap) {
    const std::vector<int>& c = g_arena.nodes[node].children;
    for (int i = 0; i < (int)c.size() && i < cap; i++) out[i] = c[i];
    return (int)c.size();
}
void rtco_node_transformation(int node, float out[16]) { mat_to(node_transformation(g_arena, node), out); }
float rtco_node_shininess(int node) { return g_arena.leaves[g_arena.nodes[node].leaf].m.shininess; }
void rtco_node_bounding_box(int node, float mn[4], float mx[4]) {
    BBox b = node_bounding_box(g_arena, node);
    put(b.min, mn);
    put(b.max, mx);
}
void rtco_node_parent_space_bounding_box(int node, float mn[4], float mx[4]) {
    BBox b = node_parent_space_bounding_box(g_arena, node);
    put(b.min, mn);
    put(b.max, mx);
}
// Shape::intersect on a node (group.rs:115-133): distances in push order + the leaf node each belongs to
int rtco_node_intersect(int node, const float o[4], const float d[4], float* ts, int* leaf_nodes, int cap) {
    std::vector<Intersection> xs;
    node_intersect(g_arena, node, ray_new(T(o), T(d)), xs);
    for (int i = 0; i < (int)xs.size() && i < cap; i++) {
        ts[i] = xs[i].distance;
        int found = -1;
        for (int k = 0; k < (int)g_arena.nodes.size(); k++)
            if (!g_arena.nodes[k].group && g_arena.nodes[k].leaf == xs[i].object) found = k;
        leaf_nodes[i] = found;
    }
    return (int)xs.size();
}
void rtco_node_world_to_object(int node, const float p[4], float out[4]) {  // shape.rs:57-61
    put(g_arena.leaves[g_arena.nodes[node].leaf].t_inverse * T(p), out);
}
void rtco_node_normal_at(int node, const float p[4], float out[4]) {
    put(normal_at(g_arena.leaves[g_arena.nodes[node].leaf], T(p)), out);
}
// World { objects: roots, light }: the arena's state is copied, so later edits do not reach this world
rtco_world* rtco_world_new_nodes(const int* roots, int n, const rtco_light* light) {
    rtco_world* w = new rtco_world();
    w->tree = g_arena;
    w->objects = g_arena.leaves;
    w->roots.assign(roots, roots + n);
    fill_light(w, light);
    return w;
}
/* ---- bounding_box.rs ---- */
void rtco_bbox_add(float mn[4], float mx[4], const float omn[4], const float omx[4]) {
    BBox b = {T(mn), T(mx)};
    bbox_add(b, {T(omn), T(omx)});
    put(b.min, mn);
    put(b.max, mx);
}
void rtco_bbox_add_point(float mn[4], float mx[4], const float p[4]) {
    BBox b = {T(mn), T(mx)};
    bbox_add_point(b, T(p));
    put(b.min, mn);
    put(b.max, mx);
}
void rtco_bbox_empty(float mn[4], float mx[4]) {
    BBox b = bbox_empty();
    put(b.min, mn);
    put(b.max, mx);
}
int rtco_bbox_contains_point(const float mn[4], const float mx[4], const float p[4]) {
    return bbox_contains_point({T(mn), T(mx)}, T(p)) ? 1 : 0;
}
int rtco_bbox_contains(const float mn[4], const float mx[4], const float omn[4], const float omx[4]) {
    return bbox_contains({T(mn), T(mx)}, {T(omn), T(omx)}) ? 1 : 0;
}
void rtco_bbox_transform(const float mn[4], const float mx[4], const float m[16], float omn[4], float omx[4]) {
    BBox b = bbox_transform({T(mn), T(mx)}, mat_from(m, 4));
    put(b.min, omn);
    put(b.max, omx);
}
void rtco_bbox_split(const float mn[4], const float mx[4], float lmn[4], float lmx[4], float rmn[4], float rmx[4]) {
    BBox l, r;
    bbox_split({T(mn), T(mx)}, &l, &r);
    put(l.min, lmn);
    put(l.max, lmx);
    put(r.min, rmn);
    put(r.max, rmx);
}
void rtco_shape_bounding_box(const rtco_shape* s, int parent_space, float mn[4], float mx[4]) {
    Shape sh = shape_from(*s, 0);
    BBox b = shape_bounding_box(sh);
    if (parent_space) b = bbox_transform(b, sh.t);
    put(b.min, mn);
    put(b.max, mx);
}

void rtco_world_free(rtco_world* w) { delete w; }
void rtco_world_set_pixel(rtco_world* w, uint32_t pixel_index) { w->pixel = pixel_index; }
uint64_t rtco_world_ray_count(const rtco_world* w) { return w->rays; }
void rtco_world_shape_inverse(const rtco_world* w, int i, float inv[16], float inv_t[16]) {
    mat_to(w->objects[i].t_inverse, inv);
    mat_to(w->objects[i].t_inverse_transpose, inv_t);
}
void rtco_light_info(const rtco_world* w, float position[4], float u_vec[4], float v_vec[4], int* cells) {
    put(w->light.position, position);
    put(w->light.u_vec, u_vec);
    put(w->light.v_vec, v_vec);
    *cells = w->light.cells;
}
int rtco_intersect(rtco_world* w, const float o[4], const float d[4], float* ts, int* objs, int cap) {
    std::vector<Intersection> xs = intersect(*w, ray_new(T(o), T(d)));
    for (int i = 0; i < (int)xs.size() && i < cap; i++) {
        ts[i] = xs[i].distance;
        objs[i] = xs[i].object;
    }
    return (int)xs.size();
}
void rtco_color_at(rtco_world* w, const float o[4], const float d[4], int depth, float out[3]) {
    putc(color_at(*w, ray_new(T(o), T(d)), depth, 1u), out);
}
int rtco_is_shadowed(rtco_world* w, const float light_pos[4], const float p[4]) {
    return is_shadowed(*w, T(light_pos), T(p)) ? 1 : 0;
}
float rtco_intensity_at(rtco_world* w, const float p[4]) { return intensity_at(*w, T(p), 1u); }
void rtco_point_on_light(rtco_world* w, int u, int v, float out[4]) { put(point_on_light(*w, u, v, 1u), out); }
void rtco_precompute(rtco_world* w, const float o[4], const float d[4], int hit, const float* ts, const int* objs,
                     int n, rtco_comps* out) {
    std::vector<Intersection> xs;
    for (int i = 0; i < n; i++) xs.push_back({ts[i], objs[i], 0.f, 0.f});
    Comps c = precompute_values(*w, ray_new(T(o), T(d)), xs[hit], xs);
    out->distance = c.distance;
    out->object = c.object;
    put(c.point, out->point);
    put(c.eye, out->eye);
    put(c.reflectv, out->reflectv);
    put(c.normal, out->normal);
    put(c.over_point, out->over_point);
    put(c.under_point, out->under_point);
    out->inside = c.inside ? 1 : 0;
    out->n1 = c.n1;
    out->n2 = c.n2;
}
void rtco_shade_hit(rtco_world* w, const rtco_comps* c, int depth, float out[3]) {
    putc(shade_hit(*w, comps_from(c), depth, 1u), out);
}
void rtco_reflected_color(rtco_world* w, const rtco_comps* c, int depth, float out[3]) {
    putc(reflected_color(*w, comps_from(c), depth, 1u), out);
}
void rtco_refracted_color(rtco_world* w, const rtco_comps* c, int depth, float out[3]) {
    putc(refracted_color(*w, comps_from(c), depth, 1u), out);
}
float rtco_schlick(const rtco_comps* c) { return schlick_reflectance(comps_from(c)); }
void rtco_phong(rtco_world* w, const rtco_material* m, const float p[4], const float eye[4], const float n[4],
                float light_intensity, float out[3]) {
    // the reference's tests pass any_shape(): an untransformed sphere (test/utils.rs)
    rtco_shape any{};
    any.kind = RTCO_SPHERE;
    for (int i = 0; i < 4; i++) any.transform[i * 5] = 1.0f;
    any.material = *m;
    Shape sh = shape_from(any, 0);
    putc(phong_lighting(sh, sh.m, w->light, T(p), T(eye), T(n), light_intensity), out);
}
void rtco_phong_on(rtco_world* w, const rtco_shape* object, const float p[4], const float eye[4], const float n[4],
                   float light_intensity, float out[3]) {
    Shape sh = shape_from(*object, 0);
    putc(phong_lighting(sh, sh.m, w->light, T(p), T(eye), T(n), light_intensity), out);
}
void rtco_pattern_color_at_world(const rtco_pattern* pat, const float p[4], float out[3]) {
    putc(pattern_color_at_world(pattern_from(*pat), T(p)), out);
}
void rtco_uv_color_at(const rtco_uv_pattern* uv, float u, float v, float out[3]) {
    rtco_pattern holder{};
    holder.kind = RTCO_PATTERN_TEXTURE_MAP;
    for (int i = 0; i < 4; i++) holder.transform[i * 5] = 1.0f;
    holder.n_uv = 1;
    holder.uv = uv;
    Pattern p = pattern_from(holder);
    putc(uv_color_at(p.uv[0], u, v), out);
}
void rtco_point_to_uv(int32_t mapping, const float p[4], float uv[2]) { point_to_uv(mapping, T(p), &uv[0], &uv[1]); }
int rtco_face_from_point(const float p[4]) { return face_from_point(T(p)); }
void rtco_cube_uv(int face, const float p[4], float uv[2]) { cube_uv(face, T(p), &uv[0], &uv[1]); }
// canvas.rs:120-197, line by line: '#' comments and blank lines dropped, magic, "w h", scale, then triplets
// that may span lines; a pixel is written for every complete triplet (write_pixel ignores out-of-canvas writes
// only loosely, canvas.rs:27 -- here extra triplets are dropped).
int rtco_canvas_from_ppm(const char* text, uint64_t len, uint32_t* w, uint32_t* h, float** rgb) {
    std::vector<std::string> lines;
    {
        std::string cur;
        for (uint64_t i = 0; i < len; i++) {
            if (text[i] == '\n') {
                lines.push_back(cur);
                cur.clear();
            } else {
                cur.push_back(text[i]);
            }
        }
        if (!cur.empty()) lines.push_back(cur);
    }
    auto trim = [](const std::string& s) {
        size_t a = 0, b = s.size();
        while (a < b && isspace((unsigned char)s[a])) a++;
        while (b > a && isspace((unsigned char)s[b - 1])) b--;
        return s.substr(a, b - a);
    };
    std::vector<std::string> clean;  // clean_line, canvas.rs:183-197
    for (const std::string& l : lines) {
        std::string t = trim(l);
        if (t.empty() || t[0] == '#') continue;
        clean.push_back(t);
    }
    auto parse_uint = [](const std::string& tok, uint64_t* out) {  // str::parse::<u32/usize>: digits, optional '+'
        size_t i = 0;
        if (!tok.empty() && tok[0] == '+') i = 1;
        if (i >= tok.size()) return false;
        uint64_t v = 0;
        for (; i < tok.size(); i++) {
            if (tok[i] < '0' || tok[i] > '9') return false;
            v = v * 10 + (uint64_t)(tok[i] - '0');
            if (v > 0xffffffffull) return false;
        }
        *out = v;
        return true;
    };
    auto split = [](const std::string& s) {
        std::vector<std::string> out;
        std::string cur;
        for (char ch : s) {
            if (isspace((unsigned char)ch)) {
                if (!cur.empty()) out.push_back(cur), cur.clear();
            } else {
                cur.push_back(ch);
            }
        }
        if (!cur.empty()) out.push_back(cur);
        return out;
    };
    if (clean.size() < 3) return 1;  // the reference unwraps and panics
    if (clean[0] != "P3") return 2;
    std::vector<std::string> dims = split(clean[1]);
    if (dims.size() != 2) return 4;
    uint64_t width, height, scale_u;
    if (!parse_uint(dims[0], &width) || !parse_uint(dims[1], &height)) return 3;
    if (!parse_uint(clean[2], &scale_u)) return 3;
    const float scale = (float)(uint32_t)scale_u;
    if (width * height > 0x0fffffffull) return 5;  // test infrastructure: no multi-GB canvases
    std::vector<float> img((size_t)width * height * 3, 0.0f);
    std::vector<uint32_t> raw;
    size_t x = 0, y = 0, head = 0;
    for (size_t li = 3; li < clean.size(); li++) {
        for (const std::string& tok : split(clean[li])) {
            uint64_t v;
            if (!parse_uint(tok, &v)) return 3;
            raw.push_back((uint32_t)v);
        }
        while (raw.size() - head >= 3) {
            float r = (float)raw[head] / scale, g = (float)raw[head + 1] / scale, b = (float)raw[head + 2] / scale;
            head += 3;
            if (!(x < width && y < height)) return 5;  // canvas.rs:27: write_pixel indexes data[y][x] and panics
            float* px = &img[(y * width + x) * 3];
            px[0] = r, px[1] = g, px[2] = b;
            x += 1;
            if (x >= width) {
                x = 0;
                y += 1;
            }
        }
    }
    *w = (uint32_t)width;
    *h = (uint32_t)height;
    *rgb = (float*)malloc(img.size() * sizeof(float) + 1);
    memcpy(*rgb, img.data(), img.size() * sizeof(float));
    return 0;
}
void rtco_pattern_color_at_object(const rtco_pattern* pat, const rtco_shape* object, const float world_point[4],
                                  float out[3]) {
    putc(pattern_color_at_object(pattern_from(*pat), shape_from(*object, 0), T(world_point)), out);
}

uint64_t rtco_render_rows(rtco_world* w, const rtco_camera* cam, int depth, int threads, uint32_t y0, uint32_t y1,
                          float* out_rgb) {
    Camera c = camera_from(cam);
    if (y1 > c.h) y1 = c.h;
    // Canvas::new is all black (canvas.rs:19-25)
    for (uint32_t y = y0; y < y1; y++) std::memset(out_rgb + (size_t)y * c.w * 3, 0, sizeof(float) * 3 * c.w);
    if (threads <= 1 || w->light.jitter_mode == RTCO_JITTER_CYCLE) {
        return render_rows_serial(*w, c, depth, y0, y1, out_rgb);
    }
    // Row-parallel CPU baseline: rows are independent (constant / hashed jitter only).
    std::atomic<uint32_t> next(y0);
    std::atomic<uint64_t> total(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) {
        pool.emplace_back([&]() {
            rtco_world local = *w;
            local.rays = 0;
            for (;;) {
                uint32_t y = next.fetch_add(1);
                if (y >= y1) break;
                render_rows_serial(local, c, depth, y, y + 1, out_rgb);
            }
            total += local.rays;
        });
    }
    for (auto& th : pool) th.join();
    w->rays += total.load();
    return total.load();
}
uint64_t rtco_render(rtco_world* w, const rtco_camera* cam, int depth, int threads, float* out_rgb) {
    return rtco_render_rows(w, cam, depth, threads, 0, cam->height, out_rgb);
}

uint8_t rtco_scale_color(float c) { return scale_color(c); }
void rtco_quantize(const float* rgb, uint64_t n, uint8_t* out) {
    for (uint64_t i = 0; i < n; i++) out[i] = scale_color(rgb[i]);
}

// canvas.rs:47-96
char* rtco_to_ppm(const float* rgb, uint32_t w, uint32_t h, uint64_t* len) {
    const size_t MAX_LINE = 70, MAX_VAL_LEN = 3;
    std::string ppm;
    ppm += "P3\n";
    ppm += std::to_string(w) + " " + std::to_string(h) + "\n";
    ppm += "255\n";
    std::string line;
    auto separator = [&]() {  // write_rgb_separator, canvas.rs:47-55
        if (line.size() < MAX_LINE - MAX_VAL_LEN) {
            line.push_back(' ');
        } else {
            ppm += line;
            ppm.push_back('\n');
            line.clear();
        }
    };
    for (uint32_t row = 0; row < h; row++) {
        line.clear();
        for (uint32_t col = 0; col < w; col++) {
            const float* p = rgb + ((size_t)row * w + col) * 3;
            line += std::to_string((unsigned)scale_color(p[0]));
            separator();
            line += std::to_string((unsigned)scale_color(p[1]));
            separator();
            line += std::to_string((unsigned)scale_color(p[2]));
            if (col != w - 1) separator();
        }
        if (!line.empty()) {
            ppm += line;
            ppm.push_back('\n');
        }
    }
    char* out = (char*)std::malloc(ppm.size() + 1);
    std::memcpy(out, ppm.data(), ppm.size());
    out[ppm.size()] = 0;
    *len = ppm.size();
    return out;
}
void rtco_free(void* p) { std::free(p); }

uint32_t rtco_jitter_hash(uint32_t seed, uint32_t pixel, uint32_t path, uint32_t cell, uint32_t draw) {
    return jitter_hash(seed, pixel, path, cell, draw);
}
float rtco_jitter_value(uint32_t h) { return jitter_value(h); }

}  // extern "C"
