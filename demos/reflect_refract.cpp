// C++ counterpart of the reference's demos/src/bin/reflect_refract.rs as shipped (its CSG object is commented
// out there): patterned reflective floor, glass ball, ringed metal ball, mirror cylinder and a cone.
//   ./reflect_refract [WIDTHxHEIGHT]   default 1000x500 (reflect_refract.rs:28-29)
#include <cstdio>
#include <iostream>

#include "rtc.hpp"
using namespace rtc;

int main(int argc, char** argv) {
    unsigned w = 1000, h = 500;
    if (argc > 1 && std::sscanf(argv[1], "%ux%u", &w, &h) != 2) return 2;
    try {
        Stripes stripes(color(1.0f, 0.2f, 0.4f), color(0.1f, 0.1f, 0.1f));
        stripes.set_transformation(scaling(0.3f, 0.3f, 0.3f) * rotation_z(3.0f * PI / 4.0f));
        Sine2D sine2d(color(0.1f, 1, 0.5f), color(0.9f, 0.2f, 0.6f));
        sine2d.set_transformation(scaling(0.05f, 1.0f, 0.05f) * translation(-5.0f, 1.0f, 0.5f));
        Plane floor = Plane::build(scaling(10.0f, 0.1f, 10.0f),
                                   Material::builder().pattern(sine2d).specular(0.0f).reflective(0.5f).build());

        Sphere middle = Sphere::build(translation(-0.5f, 1.0f, 0.5f),  // get_clear_sphere, :129-142
                                      Material::builder().color(color(0, 0, 0)).specular(1.0f).shininess(300.0f)
                                          .transparency(1.0f).refractive_index(1.52f).reflective(1.0f).build());
        middle.set_casts_shadow(false);

        Rings ring_pattern(yellow() / 2.0f, white() / 2.0f);
        ring_pattern.set_transformation(scaling(0.1f, 0.1f, 0.1f));
        Material metal_rings = metal();
        metal_rings.pattern(ring_pattern);
        Sphere right = Sphere::build(shearing(0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f) * translation(1.5f, 0.5f, -0.5f) * scaling(0.5f, 0.5f, 0.5f),
                                     metal_rings);

        Stripes stripes2 = stripes;  // much darker, since this one is also reflective (:75-78)
        stripes2.a = stripes2.a / 4.0f;
        stripes2.b = stripes2.b / 4.0f;
        Sphere left = Sphere::build(translation(-1.5f, 0.33f, -0.75f) * scaling(0.33f, 0.33f, 0.33f),
                                    Material::builder().pattern(stripes2).diffuse(0.7f).specular(1.0f).reflective(0.8f)
                                        .shininess(300.0f).build());

        Cylinder cylinder;  // get_cylinder, :144-158
        cylinder.maximum_y = 1.5f;
        cylinder.minimum_y = 0.0f;
        cylinder.set_material(Material::builder().reflective(1.0f).color(color(0.5f, 0.5f, 0.5f)).shininess(300.0f).specular(0.8f).build());
        cylinder.set_transformation(translation(3.7f, 0.0f, 4.0f) * scaling(0.33f, 1.8f, 0.33f));

        Cone cone;
        cone.maximum_y = 1.5f;
        cone.minimum_y = 0.0f;
        cone.set_material(Material::builder().color(color(0.6f, 0.3f, 0.1f)).reflective(0.5f).shininess(10.0f).specular(0.8f).build());
        cone.set_transformation(translation(-3.5f, 0.0f, 4.0f) * scaling(0.33f, 1.8f, 0.33f));

        World world;
        world.objects = {floor, left, middle, right, cylinder, cone};
        world.light = std::make_shared<PointLight>(point(-10, 10, -10), white());
        Camera camera(w, h, PI / 3.0f, view_transform(point(0, 1.5f, -5), point(0, 1, 0), vector(0, 1, 0)));
        Canvas canvas = camera.render(world, 5);
        std::cout << canvas.to_ppm() << "\n";
    } catch (const Error& e) {
        std::cerr << "reflect_refract: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
