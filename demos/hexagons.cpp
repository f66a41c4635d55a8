// C++ counterpart of the reference's demos/src/bin/hexagons.rs: a glass hexagon built from nested GroupShapes
// (six sides, each a corner sphere and an edge cylinder) in front of a checkered wall.
//   ./hexagons [WIDTHxHEIGHT]   default 1000x500 (hexagons.rs:27-28)
#include <cstdio>
#include <iostream>

#include "rtc.hpp"
using namespace rtc;

static Color from_hex(const char* code) {  // Color::from_str, color.rs:93-107
    unsigned r, g, b;
    std::sscanf(code, "#%2x%2x%2x", &r, &g, &b);
    return color((float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f);
}

static Sphere hexagon_corner(const Material& m) {
    return Sphere::build(translation(0.0f, 0.0f, -1.0f) * scaling(0.25f, 0.25f, 0.25f), m);
}

static Cylinder hexagon_edge(const Material& m) {
    Cylinder edge;
    edge.minimum_y = 0.0f;
    edge.maximum_y = 1.0f;
    edge.set_transformation(translation(0.0f, 0.0f, -1.0f) * rotation_y(-PI / 6.0f) * rotation_z(-PI / 2.0f) *
                            scaling(0.25f, 1.0f, 0.25f));
    edge.set_material(m);
    return edge;
}

static GroupShape hexagon_side(const Material& m) {
    GroupShape side;
    side.add_child(hexagon_corner(m));
    side.add_child(hexagon_edge(m));
    return side;
}

static GroupShape hexagon(const Material& m) {
    GroupShape hex;
    for (int n = 0; n <= 5; n++) {
        GroupShape side = hexagon_side(m);
        side.set_transformation(rotation_y((float)n * PI / 3.0f));
        hex.add_child(side);
    }
    return hex;
}

int main(int argc, char** argv) {
    unsigned w = 1000, h = 500;
    if (argc > 1 && std::sscanf(argv[1], "%ux%u", &w, &h) != 2) return 2;
    try {
        Plane floor;
        floor.set_transformation(translation(0.0f, 0.0f, 5.0f) * rotation_x(PI / 2.0f));
        floor.set_material(Material::builder().pattern(Checkers(from_hex("#C5D86D"), from_hex("#261C15"))).build());
        GroupShape hex1 = hexagon(glass());
        hex1.set_transformation(translation(0.0f, 0.75f, 0.0f) * rotation_x(PI / 2.0f));
        World world;
        world.objects = {floor, hex1};
        world.light = std::make_shared<PointLight>(point(-10, 10, -10), white());
        Camera camera(w, h, PI / 3.0f, view_transform(point(0, 1.5f, -5), point(0, 1, 0), vector(0, 1, 0)));
        Canvas canvas = camera.render(world, 5);
        std::cout << canvas.to_ppm() << "\n";
    } catch (const Error& e) {
        std::cerr << "hexagons: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
