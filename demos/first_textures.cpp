// C++ counterpart of the reference's demos/src/bin/first_textures.rs: planar / spherical / cylindrical / cube texture
// maps, and an image texture read from a P3 file given on the command line (the reference's demo takes its earth map
// the same way: `convert earth.jpg -compress none earth.ppm`).
//   ./first_textures EARTH.ppm [WIDTHxHEIGHT]   default 1000x500 (first_textures.rs:31-32)
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "rtc.hpp"
using namespace rtc;

int main(int argc, char** argv) {
    unsigned w = 1000, h = 500;
    if (argc < 2 || (argc > 2 && std::sscanf(argv[2], "%ux%u", &w, &h) != 2)) {
        std::cerr << "usage: first_textures EARTH.ppm [WIDTHxHEIGHT]\n";
        return 2;
    }
    try {
        std::ifstream file(argv[1]);
        if (!file) throw Error(RTC_ERR_INVALID_ARG, std::string("cannot open ") + argv[1]);
        std::stringstream text;
        text << file.rdbuf();
        Canvas earth_canvas = canvas_from_ppm(text.str());

        Plane floor = Plane::build(scaling(10.0f, 0.01f, 10.0f),
                                   Material::builder().specular(0.0f)
                                       .pattern(TextureMap(UVCheckers(16.0f, 8.0f, black(), white()), UVMapping::Planar)).build());
        Sphere sphere = Sphere::build(translation(-2.5f, 1.3f, 3.0f),
                                      Material::builder().pattern(TextureMap(UVCheckers(16.0f, 8.0f, black(), white()), UVMapping::Spherical))
                                          .diffuse(0.7f).specular(0.3f).build());
        Sphere earth = Sphere::build(translation(0.0f, 1.0f, 0.0f) * rotation_x(-0.5f) * rotation_y(-1.5f),
                                     Material::builder().pattern(TextureMap(UVImage(earth_canvas), UVMapping::Spherical))
                                         .diffuse(0.9f).specular(0.1f).shininess(10.0f).ambient(0.1f).build());
        Cylinder pedestal;  // get_pedestal, :143-160
        pedestal.maximum_y = 0.0f;
        pedestal.minimum_y = -0.15f;
        pedestal.closed = true;
        pedestal.set_material(Material::builder().color(color(0.2f, 0.2f, 0.2f)).ambient(0.0f).diffuse(0.8f).specular(0.0f).reflective(0.2f).build());
        GroupShape earth_display = GroupShape::with_children({earth, pedestal});
        earth_display.set_transformation(translation(-0.2f, 0.15f, 0.5f));

        Cylinder cylinder = Cylinder::build(translation(2.0f, 2.0f, 2.0f),
                                            Material::builder().ambient(0.1f).specular(0.6f).shininess(15.0f).diffuse(0.8f)
                                                .pattern(TextureMap(UVCheckers(16.0f, 16.0f, color(0, 0.5f, 0), white()), UVMapping::Cylindrical))
                                                .build());
        cylinder.maximum_y = 3.0f;
        cylinder.minimum_y = -3.0f;
        Cube cube;
        cube.set_transformation(translation(5.0f, 2.0f, 2.0f) * rotation_x(-PI / 4.0f));
        cube.set_material(Material::builder().pattern(align_check_cubic_map()).build());

        World world;
        world.objects = {floor, sphere, cylinder, cube, earth_display};
        // RectangleLight::new(.., None) draws its jitter from thread_rng(); the pinned hashed stream stands in for it
        world.light = std::make_shared<RectangleLight>(color(1.5f, 1.5f, 1.5f), point(-10, 10, -10), vector(2, 0, 0), 10,
                                                       vector(0, 2, 0), 10, Jitter::hashed());
        Camera camera(w, h, PI / 3.0f, view_transform(point(0, 1.5f, -10), point(2, 2.8f, 0), vector(0, 1, 0)));
        Canvas canvas = camera.render(world, 5);
        std::cout << canvas.to_ppm() << "\n";
    } catch (const Error& e) {
        std::cerr << "first_textures: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
