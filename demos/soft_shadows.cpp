// C++ counterpart of the reference's demos/src/bin/soft_shadows.rs (line-by-line: same literals, same
// object order), built on include/rtc.hpp -> librtc_amd.so.  Prints the P3 PPM to stdout exactly as
// `println!("{}", canvas.to_ppm())` does (the text plus one extra newline, soft_shadows.rs:61).
//   ./soft_shadows [WIDTHxHEIGHT]        default 1000x400 (soft_shadows.rs:24-25)
// The reference's light jitter comes from thread_rng(); here it is the pinned hashed stream.
#include <cstdio>
#include <iostream>

#include "rtc.hpp"
using namespace rtc;

static RectangleLight get_light() {  // soft_shadows.rs:72-82
    return RectangleLight(color(1.5f, 1.5f, 1.5f), point(-1, 2, 4), vector(2, 0, 0), 10, vector(0, 2, 0), 10);
}
static Cube get_lampshade() {  // :97-109
    Material m = Material::builder().color(color(1.5f, 1.5f, 1.5f)).ambient(1.f).diffuse(0.f).specular(0.f).build();
    Cube c = Cube::build(translation(0.f, 3.f, 4.f) * scaling(1.f, 1.f, 0.01f), m);
    c.set_casts_shadow(false);
    return c;
}
static Plane get_floor() {  // :117-125
    return Plane::build(identity_4x4(), Material::builder().color(white()).ambient(0.025f).diffuse(0.67f).specular(0.f).build());
}
static Sphere get_sphere_1() {  // :137-147
    return Sphere::build(translation(0.5f, 0.5f, 0.f) * scaling(0.5f, 0.5f, 0.5f),
                         Material::builder().color(red()).ambient(0.1f).specular(0.f).diffuse(0.6f).reflective(0.3f).build());
}
static Sphere get_sphere_2() {  // :159-169
    return Sphere::build(translation(-0.25f, 0.33f, 0.f) * scaling(0.33f, 0.33f, 0.33f),
                         Material::builder().color(color(0.5f, 0.5f, 1)).ambient(0.1f).specular(0.f).diffuse(0.6f).reflective(0.3f).build());
}

int main(int argc, char** argv) {
    unsigned w = 1000, h = 400;
    if (argc > 1 && std::sscanf(argv[1], "%ux%u", &w, &h) != 2) return 2;
    try {
        World world;
        world.objects = {get_lampshade(), get_floor(), get_sphere_1(), get_sphere_2()};
        world.light = std::make_shared<RectangleLight>(get_light());
        Camera camera(w, h, PI / 4.f, view_transform(point(-3, 1, 2.5f), point(0, 0.5f, 0), vector(0, 1, 0)));
        Canvas canvas = camera.render(world, 5);
        std::cerr << "Time elapsed in render() kernel: " << camera.last_stats.kernel_ms << " ms, " << camera.last_stats.rays
                  << " rays\n";
        std::cout << canvas.to_ppm() << "\n";
    } catch (const Error& e) {
        std::cerr << "soft_shadows: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
