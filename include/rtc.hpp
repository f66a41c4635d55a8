// rtc.hpp -- header-only C++17 mirror of the reference's scene API for the render hot path, on top of
// the C ABI in rtc.h.  The reference's host language is Rust and this image has no Rust toolchain, so
// this is the compiled-language host side: the same names and argument meaning as the crate
// (lib/src/{tuple,color,matrix,transformations,material,world,camera,canvas}.rs, shape/*.rs, light/*.rs),
// so a demo translates line by line (demos/*.cpp next to the reference's demos/src/bin/*.rs).
// Everything numeric happens inside librtc_amd.so; this header only marshals.
#ifndef RTC_HPP
#define RTC_HPP

#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rtc.h"

namespace rtc {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};
inline void check(int status) {
    if (status != RTC_OK) throw Error(status, rtc_last_error());
}

// ---- tuple.rs / color.rs -------------------------------------------------------------------------
struct Tuple {
    float x, y, z, w;
    const float* data() const { return &x; }
};
inline Tuple point(float x, float y, float z) { return {x, y, z, 1.0f}; }   // point!()
inline Tuple vector(float x, float y, float z) { return {x, y, z, 0.0f}; }  // vector!()
struct Color {
    float r, g, b;
    const float* data() const { return &r; }
};
inline Color color(float r, float g, float b) { return {r, g, b}; }  // color!()
inline Color white() { return {1, 1, 1}; }                           // constants.rs:19-21
inline Color black() { return {0, 0, 0}; }                           // constants.rs:22-24
inline Color red() { return {1, 0, 0}; }                             // constants.rs:25-27
inline Color yellow() { return {1, 1, 0}; }                          // constants.rs:34-36
inline Color gray() { return {0.5f, 0.5f, 0.5f}; }                   // constants.rs:43-45
inline Color operator/(Color c, float s) { return {c.r / s, c.g / s, c.b / s}; }  // color.rs:61-67

// ---- matrix.rs / transformations.rs ----------------------------------------------------------------
struct Matrix {
    float m[16];
    Matrix operator*(const Matrix& o) const {  // matrix.rs:86-103
        Matrix r;
        rtc_mat_mul(m, o.m, r.m);
        return r;
    }
    Tuple operator*(const Tuple& t) const {  // matrix.rs:73-84
        Tuple r;
        rtc_mat_vec(m, t.data(), &r.x);
        return r;
    }
    Matrix inverse() const {  // matrix.rs:201-212
        Matrix r;
        check(rtc_mat_inverse(m, 4, r.m));
        return r;
    }
};
inline Matrix identity_4x4() {
    Matrix r;
    rtc_scaling(1, 1, 1, r.m);
    return r;
}
#define RTC_HPP_MAT(name, decl, call) \
    inline Matrix name decl {         \
        Matrix r;                     \
        call;                         \
        return r;                     \
    }
RTC_HPP_MAT(translation, (float x, float y, float z), rtc_translation(x, y, z, r.m))
RTC_HPP_MAT(scaling, (float x, float y, float z), rtc_scaling(x, y, z, r.m))
RTC_HPP_MAT(rotation_x, (float a), rtc_rotation_x(a, r.m))
RTC_HPP_MAT(rotation_y, (float a), rtc_rotation_y(a, r.m))
RTC_HPP_MAT(rotation_z, (float a), rtc_rotation_z(a, r.m))
RTC_HPP_MAT(shearing, (float xy, float xz, float yx, float yz, float zx, float zy), rtc_shearing(xy, xz, yx, yz, zx, zy, r.m))
RTC_HPP_MAT(view_transform, (Tuple from, Tuple to, Tuple up), rtc_view_transform(from.data(), to.data(), up.data(), r.m))
#undef RTC_HPP_MAT

// ---- pattern/*.rs -----------------------------------------------------------------------------------
// A boxed Pattern: kind, the two colours (pub fields a / b as in stripes.rs:11-12) and the transform given to
// set_transformation (pattern.rs:20-22); the inverse is taken when the material is flattened.
// A boxed UVPattern (pattern/uv.rs:14-16): UVCheckers, AlignCheck or UVImage; an image's pixels are shared, not copied.
struct UVPattern {
    rtc_uv_pattern u{};
    std::shared_ptr<std::vector<float>> image;
};
inline UVPattern UVCheckers(float width, float height, Color a, Color b) {  // uv.rs:28-36
    UVPattern p;
    p.u.kind = RTC_UV_CHECKERS;
    p.u.width = width, p.u.height = height;
    for (int k = 0; k < 3; k++) p.u.colors[0][k] = a.data()[k], p.u.colors[1][k] = b.data()[k];
    return p;
}
inline UVPattern AlignCheck(Color main, Color ul, Color ur, Color bl, Color br) {  // uv.rs:135-143
    UVPattern p;
    p.u.kind = RTC_UV_ALIGN_CHECK;
    const Color cs[5] = {main, ul, ur, bl, br};
    for (int j = 0; j < 5; j++)
        for (int k = 0; k < 3; k++) p.u.colors[j][k] = cs[j].data()[k];
    return p;
}
enum class UVMapping : int32_t { Spherical = RTC_MAP_SPHERICAL, Planar = RTC_MAP_PLANAR, Cylindrical = RTC_MAP_CYLINDRICAL };

struct Pattern {
    int32_t kind;
    Color a, b;
    Matrix transform = identity_4x4();
    int32_t uv_mapping = 0;              // TextureMap
    std::vector<UVPattern> uv;           // TextureMap: 1; CubicMap: front, back, left, right, up, down
    mutable std::shared_ptr<std::vector<rtc_uv_pattern>> uv_c;  // what rtc_pattern::uv borrows
    Pattern(int32_t k, Color a_, Color b_) : kind(k), a(a_), b(b_) {}
    void set_transformation(Matrix t) { transform = t; }
    rtc_pattern c() const {
        rtc_pattern p;
        if (!uv.empty()) {
            uv_c = std::make_shared<std::vector<rtc_uv_pattern>>();
            for (const UVPattern& q : uv) {
                rtc_uv_pattern r = q.u;
                if (q.image) r.image_rgb = q.image->data();
                uv_c->push_back(r);
            }
            check(rtc_texture_map_init(&p, uv_mapping, uv_c->data(), (uint32_t)uv_c->size(), transform.m));
            return p;
        }
        check(rtc_pattern_init(&p, kind, a.data(), b.data(), transform.m));
        return p;
    }
};
struct TextureMap : Pattern {  // uv.rs:68-76
    TextureMap(UVPattern uv_pattern, UVMapping mapping) : Pattern(RTC_PATTERN_TEXTURE_MAP, black(), black()) {
        uv_mapping = (int32_t)mapping;
        uv.push_back(std::move(uv_pattern));
    }
};
struct CubicMap : Pattern {  // uv.rs:207-231
    CubicMap(UVPattern front, UVPattern back, UVPattern left, UVPattern right, UVPattern up, UVPattern down)
        : Pattern(RTC_PATTERN_CUBE_MAP, black(), black()) {
        uv = {std::move(front), std::move(back), std::move(left), std::move(right), std::move(up), std::move(down)};
    }
};
struct Stripes : Pattern {   // stripes.rs:17-31 (default: white, black)
    Stripes(Color a_ = white(), Color b_ = black()) : Pattern(RTC_PATTERN_STRIPES, a_, b_) {}
};
struct Gradient : Pattern {  // gradient.rs:16-23
    Gradient(Color a_ = white(), Color b_ = black()) : Pattern(RTC_PATTERN_GRADIENT, a_, b_) {}
};
struct Rings : Pattern {     // rings.rs:16-22
    Rings(Color a_ = white(), Color b_ = black()) : Pattern(RTC_PATTERN_RINGS, a_, b_) {}
};
struct Checkers : Pattern {  // checkers.rs:16-22
    Checkers(Color a_ = white(), Color b_ = black()) : Pattern(RTC_PATTERN_CHECKERS, a_, b_) {}
};
struct Sine2D : Pattern {    // sine_2d.rs:16-23
    Sine2D(Color a_ = white(), Color b_ = black()) : Pattern(RTC_PATTERN_SINE2D, a_, b_) {}
};

// ---- material.rs:18-51 (TypedBuilder defaults) -------------------------------------------------------
struct Material {
    Color color_{1, 1, 1};
    float ambient_ = 0.1f, diffuse_ = 0.9f, specular_ = 0.9f, shininess_ = 200.0f;
    float reflective_ = 0.0f, transparency_ = 0.0f, refractive_index_ = 1.0f;
    std::shared_ptr<Pattern> pattern_;  // material.rs:50 Option<BoxedPattern>
    static Material builder() { return Material(); }
    Material& pattern(const Pattern& p) { pattern_ = std::make_shared<Pattern>(p); return *this; }
    Material& color(Color c) { color_ = c; return *this; }
    Material& ambient(float v) { ambient_ = v; return *this; }
    Material& diffuse(float v) { diffuse_ = v; return *this; }
    Material& specular(float v) { specular_ = v; return *this; }
    Material& shininess(float v) { shininess_ = v; return *this; }
    Material& reflective(float v) { reflective_ = v; return *this; }
    Material& transparency(float v) { transparency_ = v; return *this; }
    Material& refractive_index(float v) { refractive_index_ = v; return *this; }
    Material build() const { return *this; }
    rtc_material c() const {
        rtc_material m;
        rtc_material_default(&m);
        m.color[0] = color_.r, m.color[1] = color_.g, m.color[2] = color_.b;
        m.ambient = ambient_, m.diffuse = diffuse_, m.specular = specular_, m.shininess = shininess_;
        m.reflective = reflective_, m.transparency = transparency_, m.refractive_index = refractive_index_;
        if (pattern_) m.pattern = pattern_->c();
        return m;
    }
};
inline Material glass() {  // constants.rs:12-17
    return Material::builder().transparency(1.0f).refractive_index(1.52f).build();
}
inline Material metal() {  // constants.rs:50-62
    return Material::builder().color(gray()).ambient(1.0f).diffuse(0.6f).reflective(0.1f).specular(0.4f).shininess(10.0f).build();
}

// ---- bounding_box.rs --------------------------------------------------------------------------------
struct BoundingBox {
    Tuple min, max;
    static BoundingBox empty() {
        BoundingBox b;
        rtc_bounds_empty(&b.min.x, &b.max.x);
        return b;
    }
    void add_bounding_box(const BoundingBox& o) { rtc_bounds_add(&min.x, &max.x, o.min.data(), o.max.data()); }
    bool contains_bounding_box(const BoundingBox& o) const {
        return rtc_bounds_contains(min.data(), max.data(), o.min.data(), o.max.data()) != 0;
    }
    std::pair<BoundingBox, BoundingBox> split() const {
        BoundingBox l, r;
        rtc_bounds_split(min.data(), max.data(), &l.min.x, &l.max.x, &r.min.x, &r.max.x);
        return {l, r};
    }
};

// ---- shape/*.rs -----------------------------------------------------------------------------------
// One value type for every Box<dyn Shape>: a leaf of some kind, or (kind == GROUP) a GroupShape holding children.
struct Shape {
    static constexpr int32_t GROUP = -1;
    int32_t kind;
    Matrix transform = identity_4x4();
    Material material;
    bool casts_shadow = true;                                   // base_shape.rs:31
    float minimum_y = -std::numeric_limits<float>::infinity();  // cylinder.rs:39-41
    float maximum_y = std::numeric_limits<float>::infinity();
    bool closed = false;
    Tuple p1{0, 0, 0, 1}, p2{0, 0, 0, 1}, p3{0, 0, 0, 1};  // Triangle (triangle.rs:11-13)
    // GroupShape state (shape/group.rs:12-16)
    std::vector<Shape> children;
    Matrix group_t_inverse = identity_4x4();   // BaseShape.t_inverse of the group itself (default: not an inversion)
    std::shared_ptr<BoundingBox> cached_box;  // cached_bounding_box: filled on first use, never invalidated

    explicit Shape(int32_t k) : kind(k) {}
    Shape(int32_t k, Matrix t, Material m) : kind(k), transform(t), material(m) {}
    bool is_group() const { return kind == GROUP; }
    const Matrix& transformation() const { return transform; }
    void set_transformation(Matrix t) {
        if (is_group()) {  // group.rs:101-114: re-bake the children
            if (!children.empty()) {
                Matrix child_transformer = t * group_t_inverse;
                for (Shape& c : children) c.set_transformation(child_transformer * c.transform);
            }
            group_t_inverse = t.inverse();
        }
        transform = t;
    }
    void set_material(Material m) {
        if (is_group()) {  // group.rs:96-100
            for (Shape& c : children) c.set_material(m);
        } else {
            material = m;
        }
    }
    void set_casts_shadow(bool v) { casts_shadow = v; }
    BoundingBox bounding_box() {  // group.rs:138-151 / the leaf kinds' bounding_box()
        BoundingBox b;
        if (kind == RTC_TRIANGLE) {
            rtc_triangle_bounds(p1.data(), p2.data(), p3.data(), nullptr, &b.min.x, &b.max.x);
            return b;
        }
        if (!is_group()) {
            check(rtc_shape_bounds(kind, minimum_y, maximum_y, nullptr, &b.min.x, &b.max.x));
            return b;
        }
        if (!cached_box) {
            b = BoundingBox::empty();
            for (Shape& c : children) b.add_bounding_box(c.parent_space_bounding_box());
            cached_box = std::make_shared<BoundingBox>(b);
        }
        return *cached_box;
    }
    BoundingBox parent_space_bounding_box() {  // shape.rs:162-164 / group.rs:153-155
        if (is_group()) return bounding_box();
        BoundingBox b;
        if (kind == RTC_TRIANGLE) rtc_triangle_bounds(p1.data(), p2.data(), p3.data(), transform.m, &b.min.x, &b.max.x);
        else check(rtc_shape_bounds(kind, minimum_y, maximum_y, transform.m, &b.min.x, &b.max.x));
        return b;
    }
    void divide(size_t threshold) {  // group.rs:157-172; a no-op for leaves (shape.rs:167)
        if (!is_group()) return;
        if (threshold <= children.size()) {
            auto halves = bounding_box().split();  // partition_children, :46-64
            std::vector<Shape> left, right, keep;
            for (Shape& c : children) {
                BoundingBox cb = c.parent_space_bounding_box();
                if (halves.first.contains_bounding_box(cb)) left.push_back(std::move(c));
                else if (halves.second.contains_bounding_box(cb)) right.push_back(std::move(c));
                else keep.push_back(std::move(c));
            }
            children = std::move(keep);
            if (!left.empty()) make_subgroup(std::move(left));
            if (!right.empty()) make_subgroup(std::move(right));
        }
        for (Shape& c : children) c.divide(threshold);
    }
    void make_subgroup(std::vector<Shape> kids) {  // group.rs:66-73
        if (kids.size() == 1) {
            children.push_back(std::move(kids[0]));
        } else {
            Shape g(GROUP);
            g.children = std::move(kids);
            children.push_back(std::move(g));
        }
    }
    rtc_object c() const {
        rtc_object o;
        rtc_material m = material.c();
        check(rtc_object_init(&o, kind, transform.m, &m));  // stores transform.inverse(), base_shape.rs:58
        o.casts_shadow = casts_shadow;
        o.closed = closed;
        o.min_y = minimum_y;
        o.max_y = maximum_y;
        for (int k = 0; k < 3; k++) o.p1[k] = p1.data()[k], o.p2[k] = p2.data()[k], o.p3[k] = p3.data()[k];
        return o;
    }
};
struct Sphere : Shape {
    Sphere() : Shape(RTC_SPHERE) {}
    static Sphere build(Matrix t, Material m) { Sphere s; s.transform = t; s.material = m; return s; }  // sphere.rs:23-28
};
struct Plane : Shape {
    Plane() : Shape(RTC_PLANE) {}
    static Plane build(Matrix t, Material m) { Plane s; s.transform = t; s.material = m; return s; }
};
struct Cube : Shape {
    Cube() : Shape(RTC_CUBE) {}
    static Cube build(Matrix t, Material m) { Cube s; s.transform = t; s.material = m; return s; }
};
struct Cylinder : Shape {
    Cylinder() : Shape(RTC_CYLINDER) {}
    static Cylinder build(Matrix t, Material m) { Cylinder s; s.transform = t; s.material = m; return s; }
};
struct Cone : Shape {  // cone.rs:12-42: minimum_y / maximum_y / closed are pub fields, as on Cylinder
    Cone() : Shape(RTC_CONE) {}
    static Cone build(Matrix t, Material m) { Cone s; s.transform = t; s.material = m; return s; }
};
struct Triangle : Shape {  // triangle.rs:19-33; e1, e2 and the normal are derived inside the library
    Triangle(Tuple a, Tuple b, Tuple c) : Shape(RTC_TRIANGLE) { p1 = a, p2 = b, p3 = c; }
};
struct SmoothTriangle : Triangle {  // smooth_triangle.rs: renders through its inner flat Triangle (:37-39)
    Tuple n1, n2, n3;
    SmoothTriangle(Tuple a, Tuple b, Tuple c, Tuple na, Tuple nb, Tuple nc) : Triangle(a, b, c), n1(na), n2(nb), n3(nc) {}
};
struct GroupShape : Shape {  // shape/group.rs
    GroupShape() : Shape(GROUP) {}
    static GroupShape with_children(std::vector<Shape> kids) { GroupShape g; g.children = std::move(kids); return g; }  // :23-27
    const std::vector<Shape>& get_children() const { return children; }
    void add_child(Shape child) {  // :39-44 (the child moves into the group, as the Box does)
        child.set_transformation(transform * child.transform);
        children.push_back(std::move(child));
    }
};

// ---- light/*.rs -----------------------------------------------------------------------------------
struct Light {
    rtc_light l;
};
struct PointLight : Light {
    PointLight(Tuple position, Color intensity) { rtc_point_light(position.data(), intensity.data(), &l); }  // point_light.rs:12-19
};
// RectangleLight::new(..., jitter_fn_opt).  `hashed(seed)` stands in for jitter_fn_opt = None (thread_rng);
// `constant(c)` for test/utils.rs constant_jitter().  An arbitrary closure cannot run on a device.
// `cycle(values)` is test/utils.rs hardcoded_jitter(): state carried from call to call, accepted by the batched
// rtc_intensity_at / rtc_point_on_light only (each point answered as by a freshly built light).
struct Jitter {
    int32_t mode;
    float value;
    uint32_t seed;
    std::vector<float> sequence;
    static Jitter hashed(uint32_t seed = 0x5EED5EEDu) { return {RTC_JITTER_HASHED, 0.0f, seed, {}}; }
    static Jitter constant(float c = 0.5f) { return {RTC_JITTER_CONSTANT, c, 0, {}}; }
    static Jitter cycle(std::vector<float> values) { return {RTC_JITTER_SEQUENCE, 0.0f, 0, std::move(values)}; }
};
struct RectangleLight : Light {
    RectangleLight(Color intensity, Tuple corner, Tuple u_vec, int32_t u_steps, Tuple v_vec, int32_t v_steps,
                   Jitter jitter = Jitter::hashed()) {  // rectangle_light.rs:33-58
        check(rtc_rectangle_light(intensity.data(), corner.data(), u_vec.data(), u_steps, v_vec.data(), v_steps,
                                  jitter.mode, jitter.value, jitter.seed, &l));
        if (jitter.mode == RTC_JITTER_SEQUENCE)
            check(rtc_light_set_jitter_sequence(&l, jitter.sequence.data(), (uint32_t)jitter.sequence.size()));
    }
};

// ---- world.rs:18-21 -------------------------------------------------------------------------------
struct World {
    std::vector<Shape> objects;
    std::shared_ptr<Light> light;  // None -> "World light should be set" (world.rs:66)
};

// ---- canvas.rs ------------------------------------------------------------------------------------
struct Canvas {
    size_t width, height;
    std::vector<float> data;  // [y][x][rgb]
    Canvas(size_t w, size_t h) : width(w), height(h), data(w * h * 3, 0.0f) {}
    Color pixel_at(size_t x, size_t y) const {
        const float* p = &data[(y * width + x) * 3];
        return {p[0], p[1], p[2]};
    }
    void write_pixel(size_t x, size_t y, Color c) {
        float* p = &data[(y * width + x) * 3];
        p[0] = c.r, p[1] = c.g, p[2] = c.b;
    }
    std::string to_ppm() const {  // canvas.rs:58-96
        char* text = nullptr;
        uint64_t len = 0;
        check(rtc_to_ppm(data.data(), (uint32_t)width, (uint32_t)height, &text, &len));
        std::string s(text, len);
        rtc_free(text);
        return s;
    }
};
inline Canvas canvas_from_ppm(const std::string& text) {  // canvas.rs:120-197
    uint32_t w = 0, h = 0;
    float* rgb = nullptr;
    check(rtc_canvas_from_ppm(text.data(), text.size(), &w, &h, &rgb));
    Canvas c(w, h);
    std::memcpy(c.data.data(), rgb, c.data.size() * sizeof(float));
    rtc_free(rgb);
    return c;
}
inline UVPattern UVImage(const Canvas& canvas) {  // uv.rs:351-355
    UVPattern p;
    p.u.kind = RTC_UV_IMAGE;
    p.u.image_width = (uint32_t)canvas.width, p.u.image_height = (uint32_t)canvas.height;
    p.image = std::make_shared<std::vector<float>>(canvas.data);
    return p;
}
inline Pattern align_check_cubic_map() {  // get_align_check_cubic_map_pattern, uv.rs:321-338
    const Color white_{1, 1, 1}, red_{1, 0, 0}, yellow_{1, 1, 0}, green_{0, 1, 0}, cyan_{0, 1, 1}, blue_{0, 0, 1},
        purple_{1, 0, 1}, brown_{1, 0.5f, 0};
    UVPattern left = AlignCheck(yellow_, cyan_, red_, blue_, brown_), front = AlignCheck(cyan_, red_, yellow_, brown_, green_);
    UVPattern right = AlignCheck(red_, yellow_, purple_, green_, white_), back = AlignCheck(green_, purple_, cyan_, white_, blue_);
    UVPattern up = AlignCheck(brown_, cyan_, purple_, red_, yellow_), down = AlignCheck(purple_, brown_, green_, blue_, white_);
    return CubicMap(front, back, left, right, up, down);
}

// ---- camera.rs ------------------------------------------------------------------------------------
struct Camera {
    rtc_camera c;
    rtc_stats last_stats{};
    // depth-first leaves + one rtc_group (leaf run, bounding box) per GroupShape -- see rtc_group in rtc.h
    static void flatten(Shape& s, std::vector<rtc_object>& objs, std::vector<rtc_group>& groups) {
        if (!s.is_group()) {
            objs.push_back(s.c());
            return;
        }
        const size_t gi = groups.size();
        BoundingBox b = s.bounding_box();
        groups.push_back({(uint32_t)objs.size(), 0u, {b.min.x, b.min.y, b.min.z}, {b.max.x, b.max.y, b.max.z}});
        for (Shape& c : s.children) flatten(c, objs, groups);
        groups[gi].n_objects = (uint32_t)objs.size() - groups[gi].first_object;
    }
    Camera(uint32_t width_pixels, uint32_t height_pixels, float field_of_view, Matrix transform) {  // camera.rs:23-56
        check(rtc_camera_new(width_pixels, height_pixels, field_of_view, transform.m, &c));
    }
    // Camera::render (camera.rs:76-91) on the MI355X
    Canvas render(World world, int16_t reflection_recursion_depth, int device = 0) {  // takes the World by value, as the reference does
        std::vector<rtc_object> objs;
        std::vector<rtc_group> groups;
        for (Shape& s : world.objects) flatten(s, objs, groups);
        rtc_scene scene{(uint32_t)objs.size(), objs.data(), world.light ? &world.light->l : nullptr,
                        (uint32_t)groups.size(), groups.data()};
        Canvas canvas(c.width, c.height);
        check(rtc_render(&scene, &c, reflection_recursion_depth, device, canvas.data.data(), &last_stats));
        return canvas;
    }
    // the same frame split over several GPUs (rtc_render_ex: the image's 64-row bands dealt round-robin over `devices`)
    Canvas render(World world, int16_t reflection_recursion_depth, const std::vector<int32_t>& devices) {
        std::vector<rtc_object> objs;
        std::vector<rtc_group> groups;
        for (Shape& s : world.objects) flatten(s, objs, groups);
        rtc_scene scene{(uint32_t)objs.size(), objs.data(), world.light ? &world.light->l : nullptr,
                        (uint32_t)groups.size(), groups.data()};
        rtc_opts opts{devices.data(), (uint32_t)devices.size(), 0u, 0, 0};
        Canvas canvas(c.width, c.height);
        check(rtc_render_ex(&scene, &c, reflection_recursion_depth, &opts, canvas.data.data(), &last_stats));
        return canvas;
    }
};

constexpr float PI = 3.14159265358979323846264338327950288f;  // std::f32::consts::PI

}  // namespace rtc
#endif
