// C++ counterpart of the reference's demos/src/bin/first_patterns.rs: a Sine2D floor and three striped spheres.
//   ./first_patterns [WIDTHxHEIGHT]   default 100x50 (first_patterns.rs:25-26)
#include <cstdio>
#include <iostream>

#include "rtc.hpp"
using namespace rtc;

int main(int argc, char** argv) {
    unsigned w = 100, h = 50;
    if (argc > 1 && std::sscanf(argv[1], "%ux%u", &w, &h) != 2) return 2;
    try {
        Stripes stripes(color(1.0f, 0.2f, 0.4f), color(0.1f, 0.1f, 0.1f));
        stripes.set_transformation(scaling(0.3f, 0.3f, 0.3f) * rotation_z(3.0f * PI / 4.0f));
        Sine2D sine2d(color(0.1f, 1, 0.5f), color(0.9f, 0.2f, 0.6f));
        sine2d.set_transformation(scaling(0.005f, 1.0f, 0.005f) * translation(-5.0f, 1.0f, 0.5f));
        Plane floor = Plane::build(scaling(10.0f, 0.01f, 10.0f), Material::builder().pattern(sine2d).specular(0.0f).build());
        Material striped = Material::builder().pattern(stripes).diffuse(0.7f).specular(0.3f).build();
        Sphere middle = Sphere::build(translation(-0.5f, 1.0f, 0.5f), striped);
        Sphere right = Sphere::build(shearing(0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f) * translation(1.5f, 0.5f, -0.5f) * scaling(0.5f, 0.5f, 0.5f),
                                     striped);
        Sphere left = Sphere::build(translation(-1.5f, 0.33f, -0.75f) * scaling(0.33f, 0.33f, 0.33f), striped);
        World world;
        world.objects = {floor, left, middle, right};
        world.light = std::make_shared<PointLight>(point(-10, 10, -10), white());
        Camera camera(w, h, PI / 3.0f, view_transform(point(0, 1.5f, -5), point(0, 1, 0), vector(0, 1, 0)));
        Canvas canvas = camera.render(world, 5);
        std::cout << canvas.to_ppm() << "\n";
    } catch (const Error& e) {
        std::cerr << "first_patterns: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
