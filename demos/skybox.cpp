// C++ counterpart of the reference's demos/src/bin/skybox.rs: a reflective sphere inside a cube whose six faces are image
// textures (CubicMap of UVImages), read from P3 files in the directory given on the command line -- the reference's demo takes
// them the same way (posz / negz / posx / negx / posy / negy .ppm, `convert x.jpg -compress none x.ppm`).
//   ./skybox DIRECTORY [WIDTHxHEIGHT]   default 800x400 (skybox.rs:23-24)
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "rtc.hpp"
using namespace rtc;

static Canvas face(const std::string& dir, const char* name) {  // get_uv_from_path, skybox.rs:127-131
    const std::string path = dir + "/" + name;
    std::ifstream file(path);
    if (!file) throw Error(RTC_ERR_INVALID_ARG, "cannot open " + path);
    std::stringstream text;
    text << file.rdbuf();
    return canvas_from_ppm(text.str());
}

int main(int argc, char** argv) {
    unsigned w = 800, h = 400;
    if (argc < 2 || (argc > 2 && std::sscanf(argv[2], "%ux%u", &w, &h) != 2)) {
        std::cerr << "usage: skybox DIRECTORY [WIDTHxHEIGHT]\n";
        return 2;
    }
    try {
        const std::string dir = argv[1];
        Sphere sphere = Sphere::build(scaling(0.75f, 0.75f, 0.75f) * translation(0.0f, 0.0f, 5.0f),  // :43-55
                                      Material::builder().diffuse(0.4f).specular(0.6f).shininess(20.0f).reflective(0.6f).ambient(0.0f).build());
        // :82-93 -- the demo hands posx.ppm to `left` and negx.ppm to `right`, as written there
        Canvas front = face(dir, "posz.ppm"), back = face(dir, "negz.ppm"), left = face(dir, "posx.ppm"), right = face(dir, "negx.ppm"),
               up = face(dir, "posy.ppm"), down = face(dir, "negy.ppm");
        Cube sky = Cube::build(scaling(1000.0f, 1000.0f, 1000.0f),  // :95-102
                               Material::builder().diffuse(0.0f).specular(0.0f).ambient(1.0f)
                                   .pattern(CubicMap(UVImage(front), UVImage(back), UVImage(left), UVImage(right), UVImage(up), UVImage(down))).build());
        World world;
        world.objects = {sphere, sky};
        world.light = std::make_shared<PointLight>(point(0, 100, 0), color(1, 1, 1));  // get_light, :123-125
        Camera camera(w, h, 1.2f, view_transform(point(0, 0, 0), point(0, 0, 5), vector(0, 1, 0)));
        Canvas canvas = camera.render(world, 5);
        std::cout << canvas.to_ppm() << "\n";
    } catch (const Error& e) {
        std::cerr << "skybox: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
