"""Scene scripts for the BASELINE.json configurations and the in-scope demos.

Each function returns (world, camera, depth) built with the mirror API of
api.py -- i.e. exactly what the corresponding demos/src/bin/*.rs file does before
it calls camera.render(world, depth).  Literals are the reference's.
"""
import numpy as np

from .api import (Camera, Checkers, Cone, Cube, Cylinder, Gradient, Material, Plane, PointLight, RectangleLight, Rings,
                  Sine2D, Sphere, Stripes, World, chain, color, identity_4x4, metal, point, rotation_x, rotation_y,
                  rotation_z, scaling, shearing, translation, vector, view_transform)

f32 = np.float32
PI = f32(3.14159265358979323846264338327950288)  # std::f32::consts::PI

DEFAULT_SEED = 0x5EED5EED


def soft_shadows(width=1000, height=400, jitter=("hashed", DEFAULT_SEED)):
    """demos/src/bin/soft_shadows.rs:33-169 (C1: 1000x400, C3: 4096x4096)."""
    light = RectangleLight(color(1.5, 1.5, 1.5), point(-1, 2, 4), vector(2, 0, 0), 10, vector(0, 2, 0), 10, jitter)
    lampshade = Cube(chain(translation(0.0, 3.0, 4.0), scaling(1.0, 1.0, 0.01)),
                     Material(color=(1.5, 1.5, 1.5), ambient=1.0, diffuse=0.0, specular=0.0), casts_shadow=False)
    floor = Plane(identity_4x4(), Material(color=(1, 1, 1), ambient=0.025, diffuse=0.67, specular=0.0))
    sphere_1 = Sphere(chain(translation(0.5, 0.5, 0.0), scaling(0.5, 0.5, 0.5)),
                      Material(color=(1, 0, 0), ambient=0.1, specular=0.0, diffuse=0.6, reflective=0.3))
    sphere_2 = Sphere(chain(translation(-0.25, 0.33, 0.0), scaling(0.33, 0.33, 0.33)),
                      Material(color=(0.5, 0.5, 1), ambient=0.1, specular=0.0, diffuse=0.6, reflective=0.3))
    world = World([lampshade, floor, sphere_1, sphere_2], light)
    camera = Camera(width, height, PI / f32(4.0),
                    view_transform(point(-3, 1, 2.5), point(0, 0.5, 0), vector(0, 1, 0)))
    return world, camera, 5


def single_sphere(width=1024, height=1024):
    """C2: one default sphere + point light, primary and shadow rays only (world.rs:32-48 family)."""
    world = World([Sphere(identity_4x4(), Material())], PointLight(point(-10, 10, -10), color(1, 1, 1)))
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0, 0, -5), point(0, 0, 0), vector(0, 1, 0)))
    return world, camera, 5


def glass_and_mirror(width=4096, height=4096):
    """C4: the demo's glass ball (reflect_refract.rs:129-142, casting shadows) over a mirror plane."""
    ball = Sphere(translation(0.0, 1.0, 0.0),
                  Material(color=(0, 0, 0), specular=1.0, shininess=300.0, transparency=1.0, refractive_index=1.52,
                           reflective=1.0))
    floor = Plane(identity_4x4(), Material(color=(0.8, 0.8, 0.8), specular=0.0, reflective=0.8))
    world = World([ball, floor], PointLight(point(-10, 10, -10), color(1, 1, 1)))
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0, 1.5, -5), point(0, 1, 0), vector(0, 1, 0)))
    return world, camera, 5


def _xorshift32(state):
    state ^= (state << 13) & 0xFFFFFFFF
    state ^= state >> 17
    state ^= (state << 5) & 0xFFFFFFFF
    return state & 0xFFFFFFFF


def sphere_grid(width=8192, height=8192, n=8):
    """C5: n*n unit spheres scaled 0.4 on a grid, colours from xorshift32."""
    objects = []
    s = 0x9E3779B9
    for i in range(n):
        for j in range(n):
            rgb = []
            for _ in range(3):
                s = _xorshift32(s)
                rgb.append(float(f32(0.2) + f32(0.8) * f32(s / 4294967296.0)))
            m = Material(color=tuple(rgb), reflective=0.3 if (i + j) % 2 else 0.0)
            objects.append(Sphere(chain(translation(-7.0 + 2.0 * i, 0.4, 2.0 * j), scaling(0.4, 0.4, 0.4)), m))
    world = World(objects, PointLight(point(-10, 20, -10), color(1, 1, 1)))
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0, 12, -14), point(0, 0, 7), vector(0, 1, 0)))
    return world, camera, 5


def first_scene(width=1000, height=500):
    """demos/src/bin/first_scene.rs:26-107 (six spheres incl. sheared / rotated ones)."""
    room = Material(color=(1, 0.9, 0.9), specular=0.0)
    floor = Sphere(scaling(10.0, 0.01, 10.0), room)
    left_wall = Sphere(chain(translation(0.0, 0.0, 5.0), rotation_y(-PI / f32(4.0)), rotation_x(PI / f32(2.0)),
                             scaling(10.0, 0.01, 10.0)), room)
    right_wall = Sphere(chain(translation(0.0, 0.0, 5.0), rotation_y(PI / f32(4.0)), rotation_x(PI / f32(2.0)),
                              scaling(10.0, 0.01, 10.0)), room)
    middle = Sphere(translation(-0.5, 1.0, 0.5), Material(color=(0.1, 1, 0.5), diffuse=0.7, specular=0.3))
    right = Sphere(chain(shearing(0.0, 1.0, 0.0, 0.0, 0.0, 1.0), translation(1.5, 0.5, -0.5), scaling(0.5, 0.5, 0.5)),
                   Material(color=(0.5, 1, 0.1), diffuse=0.7, specular=0.3))
    left = Sphere(chain(translation(-1.5, 0.33, -0.75), scaling(0.33, 0.33, 0.33)),
                  Material(color=(1, 0.8, 0.1), diffuse=0.7, specular=0.3))
    world = World([floor, left_wall, right_wall, left, middle, right], PointLight(point(-10, 10, -10), color(1, 1, 1)))
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0, 1.5, -5), point(0, 1, 0), vector(0, 1, 0)))
    return world, camera, 5


def first_plane(width=100, height=50):
    """demos/src/bin/first_plane.rs:23-86."""
    floor = Plane(scaling(10.0, 0.01, 10.0), Material(color=(1, 0.9, 0.9), specular=0.0))
    middle = Sphere(translation(-0.5, 1.0, 0.5), Material(color=(0.1, 1, 0.5), diffuse=0.7, specular=0.3))
    right = Sphere(chain(shearing(0.0, 1.0, 0.0, 0.0, 0.0, 1.0), translation(1.5, 0.5, -0.5), scaling(0.5, 0.5, 0.5)),
                   Material(color=(0.5, 1, 0.1), diffuse=0.7, specular=0.3))
    left = Sphere(chain(translation(-1.5, 0.33, -0.75), scaling(0.33, 0.33, 0.33)),
                  Material(color=(1, 0.8, 0.1), diffuse=0.7, specular=0.3))
    world = World([floor, left, middle, right], PointLight(point(-10, 10, -10), color(1, 1, 1)))
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0, 1.5, -5), point(0, 1, 0), vector(0, 1, 0)))
    return world, camera, 5


def _demo_stripes():
    """first_patterns.rs:29-30 / reflect_refract.rs:37-38"""
    return Stripes((1.0, 0.2, 0.4), (0.1, 0.1, 0.1),
                   chain(scaling(0.3, 0.3, 0.3), rotation_z(f32(3.0) * PI / f32(4.0))))


def first_patterns(width=100, height=50):
    """demos/src/bin/first_patterns.rs:28-93 (default 100x50; the commented-out size is 1000x500)."""
    stripes = _demo_stripes()
    sine2d = Sine2D((0.1, 1, 0.5), (0.9, 0.2, 0.6), chain(scaling(0.005, 1.0, 0.005), translation(-5.0, 1.0, 0.5)))
    floor = Plane(scaling(10.0, 0.01, 10.0), Material(pattern=sine2d, specular=0.0))
    middle = Sphere(translation(-0.5, 1.0, 0.5), Material(pattern=stripes, diffuse=0.7, specular=0.3))
    right = Sphere(chain(shearing(0.0, 1.0, 0.0, 0.0, 0.0, 1.0), translation(1.5, 0.5, -0.5), scaling(0.5, 0.5, 0.5)),
                   Material(pattern=stripes, diffuse=0.7, specular=0.3))
    left = Sphere(chain(translation(-1.5, 0.33, -0.75), scaling(0.33, 0.33, 0.33)),
                  Material(pattern=stripes, diffuse=0.7, specular=0.3))
    world = World([floor, left, middle, right], PointLight(point(-10, 10, -10), color(1, 1, 1)))
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0, 1.5, -5), point(0, 1, 0), vector(0, 1, 0)))
    return world, camera, 5


def reflect_refract(width=1000, height=500):
    """demos/src/bin/reflect_refract.rs:35-158 as shipped (the CSG object is commented out there, :107)."""
    stripes = _demo_stripes()
    sine2d = Sine2D((0.1, 1, 0.5), (0.9, 0.2, 0.6), chain(scaling(0.05, 1.0, 0.05), translation(-5.0, 1.0, 0.5)))
    floor = Plane(scaling(10.0, 0.1, 10.0), Material(pattern=sine2d, specular=0.0, reflective=0.5))
    middle = Sphere(translation(-0.5, 1.0, 0.5),
                    Material(color=(0, 0, 0), specular=1.0, shininess=300.0, transparency=1.0, refractive_index=1.52,
                             reflective=1.0), casts_shadow=False)
    half = f32(2.0)
    ring_pattern = Rings(tuple(f32(c) / half for c in (1, 1, 0)), tuple(f32(c) / half for c in (1, 1, 1)),
                         scaling(0.1, 0.1, 0.1))  # yellow() / 2., white() / 2.
    right = Sphere(chain(shearing(0.0, 1.0, 0.0, 0.0, 0.0, 1.0), translation(1.5, 0.5, -0.5), scaling(0.5, 0.5, 0.5)),
                   metal().copy(pattern=ring_pattern))
    quarter = f32(4.0)
    stripes2 = Stripes(tuple(f32(c) / quarter for c in stripes.a), tuple(f32(c) / quarter for c in stripes.b),
                       stripes.transform)
    left = Sphere(chain(translation(-1.5, 0.33, -0.75), scaling(0.33, 0.33, 0.33)),
                  Material(pattern=stripes2, diffuse=0.7, specular=1.0, reflective=0.8, shininess=300.0))
    cylinder = Cylinder(chain(translation(3.7, 0.0, 4.0), scaling(0.33, 1.8, 0.33)),
                        Material(reflective=1.0, color=(0.5, 0.5, 0.5), shininess=300.0, specular=0.8),
                        minimum_y=0.0, maximum_y=1.5)
    cone = Cone(chain(translation(-3.5, 0.0, 4.0), scaling(0.33, 1.8, 0.33)),
                Material(color=(0.6, 0.3, 0.1), reflective=0.5, shininess=10.0, specular=0.8),
                minimum_y=0.0, maximum_y=1.5)
    world = World([floor, left, middle, right, cylinder, cone], PointLight(point(-10, 10, -10), color(1, 1, 1)))
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0, 1.5, -5), point(0, 1, 0), vector(0, 1, 0)))
    return world, camera, 5


def patterns_medley(width=256, height=192, jitter=("hashed", 11)):
    """All five patterns and open / closed / double cones under an area light: a parity stress scene
    (not a reference demo)."""
    floor = Plane(identity_4x4(), Material(pattern=Checkers((0.9, 0.9, 0.9), (0.15, 0.15, 0.2),
                                                          chain(rotation_y(f32(0.3)), scaling(0.7, 0.7, 0.7))),
                                           specular=0.1, reflective=0.2))
    wall = Plane(chain(translation(0.0, 0.0, 6.0), rotation_x(PI / f32(2.0))),
                 Material(pattern=Sine2D((0.1, 0.3, 0.8), (0.9, 0.8, 0.2), scaling(0.07, 1.0, 0.07)), specular=0.0))
    closed_cone = Cone(chain(translation(-1.8, 1.2, 0.5), scaling(0.6, 1.2, 0.6)),
                       Material(pattern=Gradient((1.0, 0.1, 0.1), (0.1, 0.1, 1.0),
                                                 chain(translation(-1.0, 0.0, 0.0), scaling(2.0, 1.0, 1.0))),
                                diffuse=0.8, specular=0.4, shininess=40.0),
                       minimum_y=-1.0, maximum_y=0.0, closed=True)
    hourglass = Cone(chain(translation(1.9, 1.0, 1.0), rotation_z(f32(0.2)), scaling(0.5, 1.0, 0.5)),
                     Material(color=(0.1, 0.1, 0.1), transparency=0.8, refractive_index=1.333, reflective=0.4,
                              diffuse=0.3),
                     minimum_y=-1.0, maximum_y=1.0, closed=True)
    ringed = Sphere(chain(translation(0.0, 0.8, -0.5), scaling(0.8, 0.8, 0.8)),
                    Material(pattern=Rings((0.9, 0.7, 0.1), (0.2, 0.5, 0.3),
                                           chain(rotation_x(f32(0.9)), scaling(0.15, 0.15, 0.15))),
                             diffuse=0.7, specular=0.6, reflective=0.15))
    striped_box = Cube(chain(translation(-0.4, 0.3, -2.2), rotation_y(f32(0.7)), scaling(0.3, 0.3, 0.3)),
                       Material(pattern=_demo_stripes(), ambient=0.2))
    light = RectangleLight(color(1.3, 1.3, 1.3), point(-3, 5, -5), vector(2, 0, 0), 3, vector(0, 0, 2), 3, jitter)
    world = World([floor, wall, closed_cone, hourglass, ringed, striped_box], light)
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0.3, 2.4, -6.5), point(0, 0.9, 0), vector(0, 1, 0)))
    return world, camera, 5


def _hex_color(code):
    """Color::from_str, color.rs:93-107: '#rrggbb' -> u8 / 255.0 per channel"""
    return tuple(f32(int(code[i:i + 2], 16)) / f32(255.0) for i in (1, 3, 5))


def hexagon(api, material):
    """demos/src/bin/hexagons.rs:67-105: six sides, each a group of a corner sphere and an edge cylinder.
    `api` is the module providing GroupShape / Sphere / Cylinder and the transforms (this package, or the oracle's
    mirror of the same names in tests -- both then run their own transform-baking code)."""
    hexa = api.GroupShape()
    for n in range(6):
        side = api.GroupShape()
        side.add_child(api.Sphere(api.chain(api.translation(0.0, 0.0, -1.0), api.scaling(0.25, 0.25, 0.25)), material))
        side.add_child(api.Cylinder(api.chain(api.translation(0.0, 0.0, -1.0), api.rotation_y(-PI / f32(6.0)),
                                              api.rotation_z(-PI / f32(2.0)), api.scaling(0.25, 1.0, 0.25)),
                                    material, minimum_y=0.0, maximum_y=1.0))
        side.set_transformation(api.rotation_y(f32(n) * PI / f32(3.0)))
        hexa.add_child(side)
    return hexa


def hexagons_objects(api):
    """World.objects of demos/src/bin/hexagons.rs:32-51: a checkered wall and a glass hexagon."""
    floor = api.Plane(api.chain(api.translation(0.0, 0.0, 5.0), api.rotation_x(PI / f32(2.0))),
                      api.Material(pattern=api.Checkers(_hex_color("#C5D86D"), _hex_color("#261C15"))))
    hex1 = hexagon(api, api.Material(transparency=1.0, refractive_index=1.52))  # constants.rs glass()
    hex1.set_transformation(api.chain(api.translation(0.0, 0.75, 0.0), api.rotation_x(PI / f32(2.0))))
    return [floor, hex1]


def hexagons(width=1000, height=500):
    """demos/src/bin/hexagons.rs:32-65."""
    from . import api
    world = World(hexagons_objects(api), PointLight(point(-10, 10, -10), color(1, 1, 1)))
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0, 1.5, -5), point(0, 1, 0), vector(0, 1, 0)))
    return world, camera, 5


def grouped_grid_objects(api, n=8, threshold=4):
    """C5's n*n sphere grid held in ONE GroupShape and subdivided with divide(threshold) (group.rs:157-172):
    the reference's own bounding-volume hierarchy over the same spheres."""
    grid = api.GroupShape()
    s = 0x9E3779B9
    for j in range(n):
        for i in range(n):
            rgb = []
            for _ in range(3):
                s = _xorshift32(s)
                rgb.append(f32(s >> 8) / f32(1 << 24))
            reflective = 0.3 if (i + j) % 2 == 0 else 0.0
            grid.add_child(api.Sphere(api.chain(api.translation(f32(i) - f32(n - 1) / f32(2.0), 0.4, f32(j) - f32(n - 1) / f32(2.0)),
                                                api.scaling(0.4, 0.4, 0.4)),
                                      api.Material(color=tuple(rgb), diffuse=0.7, specular=0.3, reflective=reflective)))
    grid.divide(threshold)
    floor = api.Plane(api.identity_4x4(), api.Material(color=(0.9, 0.9, 0.9), specular=0.0, reflective=0.1))
    return [floor, grid]


def grouped_grid(width=8192, height=8192, n=8, threshold=4):
    from . import api
    world = World(grouped_grid_objects(api, n, threshold), PointLight(point(-10, 10, -10), color(1, 1, 1)))
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0, 6.0, -9.0), point(0, 0, 0), vector(0, 1, 0)))
    return world, camera, 5


def groups_medley_objects(api):
    """Nested, transformed and subdivided groups of every shape kind with patterns, a non-casting leaf, an empty
    group and a plane inside a group (infinite bounds): a parity stress scene (not a reference demo)."""
    inner = api.GroupShape()
    inner.set_transformation(api.scaling(0.6, 0.6, 0.6))
    inner.add_child(api.Cube(api.translation(0.0, 1.0, 0.0), api.Material(color=(1.0, 0.5, 0.1), reflective=0.2)))
    inner.add_child(api.Cone(api.chain(api.translation(0.0, 3.2, 0.0), api.scaling(0.7, 1.2, 0.7)),
                             api.Material(pattern=api.Gradient((0.9, 0.1, 0.1), (0.1, 0.1, 0.9))),
                             minimum_y=-1.0, maximum_y=0.0, closed=True))
    outer = api.GroupShape()
    outer.add_child(api.Sphere(api.translation(-2.0, 1.0, 0.0),
                               api.Material(color=(0.05, 0.05, 0.05), transparency=0.9, refractive_index=1.52, reflective=0.6)))
    outer.add_child(inner)
    outer.add_child(api.Cylinder(api.chain(api.translation(2.2, 0.0, 0.5), api.scaling(0.5, 1.0, 0.5)),
                                 api.Material(pattern=api.Stripes((0.2, 0.8, 0.3), (0.9, 0.9, 0.9), api.scaling(0.25, 0.25, 0.25))),
                                 minimum_y=0.0, maximum_y=2.0, closed=True))
    outer.add_child(api.GroupShape())  # empty group
    outer.set_transformation(api.chain(api.translation(0.3, 0.0, 1.0), api.rotation_y(f32(0.4))))
    swarm = api.GroupShape()
    s = 12345
    for k in range(24):
        s = _xorshift32(s)
        x = f32(s & 0xFFFF) / f32(65536.0) * f32(8.0) - f32(4.0)
        s = _xorshift32(s)
        z = f32(s & 0xFFFF) / f32(65536.0) * f32(6.0) - f32(4.0)
        s = _xorshift32(s)
        y = f32(s & 0xFFFF) / f32(65536.0) * f32(2.5) + f32(0.2)
        swarm.add_child(api.Sphere(api.chain(api.translation(x, y, z), api.scaling(0.18, 0.18, 0.18)),
                                   api.Material(color=(0.3 + 0.03 * k, 0.9 - 0.03 * k, 0.5), reflective=0.1 * (k % 3)),
                                   casts_shadow=(k % 5 != 0)))
    swarm.divide(3)
    ground = api.GroupShape()
    ground.add_child(api.Plane(api.identity_4x4(), api.Material(pattern=api.Checkers((0.8, 0.8, 0.8), (0.3, 0.3, 0.35)),
                                                               reflective=0.15)))
    return [ground, outer, swarm]


def groups_medley(width=256, height=192, jitter=("hashed", 3)):
    from . import api
    light = RectangleLight(color(1.2, 1.2, 1.2), point(-4, 6, -5), vector(2, 0, 0), 3, vector(0, 0, 2), 2, jitter)
    world = World(groups_medley_objects(api), light)
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0.5, 3.0, -7.5), point(0, 1.0, 0), vector(0, 1, 0)))
    return world, camera, 5


def bumpy_mesh_obj(nu=16, nv=10, with_normals=False):
    """A closed, bumpy UV-sphere as Wavefront OBJ text (quads + two triangle fans), the kind of input the
    reference's `here_be_dragons` demo reads from a file that is not in its repository."""
    import math
    lines = ["# procedural test mesh: %d x %d" % (nu, nv)]
    verts = []
    for j in range(1, nv):
        th = math.pi * j / nv
        for i in range(nu):
            ph = 2.0 * math.pi * i / nu
            r = 1.0 + 0.15 * math.sin(3 * ph) * math.sin(2 * th)
            verts.append((r * math.sin(th) * math.cos(ph), r * math.cos(th), r * math.sin(th) * math.sin(ph)))
    verts.append((0.0, 1.0, 0.0))
    verts.append((0.0, -1.0, 0.0))
    for v in verts:
        lines.append("v %.5f %.5f %.5f" % v)
    if with_normals:
        for v in verts:
            n = math.sqrt(sum(c * c for c in v))
            lines.append("vn %.4f %.4f %.4f" % tuple(c / n for c in v))
    top, bottom = len(verts) - 1, len(verts)

    def ref(k):
        return "%d//%d" % (k, k) if with_normals else "%d" % k
    lines.append("g body")
    for j in range(nv - 2):
        for i in range(nu):
            a = j * nu + i + 1
            b = j * nu + (i + 1) % nu + 1
            lines.append("f %s %s %s %s" % (ref(a), ref(b), ref(b + nu), ref(a + nu)))
    lines.append("g caps")
    for i in range(nu):
        a, b = i + 1, (i + 1) % nu + 1
        lines.append("f %s %s %s" % (ref(top), ref(b), ref(a)))
        a2, b2 = (nv - 2) * nu + i + 1, (nv - 2) * nu + (i + 1) % nu + 1
        lines.append("f %s %s %s" % (ref(bottom), ref(a2), ref(b2)))
    return "\n".join(lines) + "\n"


def mesh_objects(api, nu=16, nv=10, threshold=6):
    """Two parsed OBJ meshes (one flat-shaded, one with vertex normals) as divided GroupShapes, a loose
    top-level triangle and a floor."""
    from .obj_parser import parse_obj
    flat = parse_obj(bumpy_mesh_obj(nu, nv, False), api).take_all_as_group()
    flat.set_material(api.Material(color=(0.9, 0.4, 0.2), diffuse=0.7, specular=0.4, shininess=60.0, reflective=0.1))
    flat.set_transformation(api.chain(api.translation(-1.3, 1.0, 0.0), api.rotation_y(f32(0.5))))
    flat.divide(threshold)
    smooth = parse_obj(bumpy_mesh_obj(nu, nv, True), api).take_all_as_group()
    smooth.set_material(api.Material(color=(0.1, 0.1, 0.15), transparency=0.85, refractive_index=1.52, reflective=0.5,
                                     diffuse=0.3))
    smooth.set_transformation(api.chain(api.translation(1.4, 0.8, 0.4), api.scaling(0.8, 0.8, 0.8)))
    smooth.divide(threshold)
    sail = api.Triangle(point(-0.5, 0.0, 2.5), point(1.0, 0.0, 2.2), point(0.2, 2.6, 2.4), None,
                        api.Material(pattern=api.Stripes((0.2, 0.5, 0.9), (0.9, 0.9, 0.9), api.scaling(0.2, 0.2, 0.2))))
    floor = api.Plane(api.identity_4x4(), api.Material(color=(0.8, 0.8, 0.75), specular=0.0, reflective=0.2))
    return [floor, flat, smooth, sail]


def mesh(width=512, height=384, nu=16, nv=10, threshold=6):
    from . import api
    world = World(mesh_objects(api, nu, nv, threshold), PointLight(point(-6, 8, -8), color(1, 1, 1)))
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0.2, 2.2, -5.5), point(0, 0.8, 0), vector(0, 1, 0)))
    return world, camera, 5


def dragon_stand_in_obj(nu=48, nv=32):
    """Wavefront OBJ text of a closed, ridged, flattened blob whose normalised bounds are close to those the demo
    notes for its dragon (x in [-1, 1], |y| <= 0.69, |z| <= 0.44; here_be_dragons.rs:296-298): stands in for the
    `dragon.obj` the reference's demo reads from a path that is not in its repository.  nu * (nv - 2) quads (fan-
    triangulated by the parser) + 2 * nu cap triangles."""
    import math
    lines = ["# stand-in for dragon.obj: %d x %d" % (nu, nv)]
    verts = []
    for j in range(1, nv):
        th = math.pi * j / nv
        for i in range(nu):
            ph = 2.0 * math.pi * i / nu
            r = 1.0 + 0.12 * math.sin(5 * ph) * math.sin(3 * th) + 0.05 * math.cos(9 * ph + 2 * th)
            verts.append((r * math.sin(th) * math.cos(ph), 0.69 * r * math.cos(th), 0.44 * r * math.sin(th) * math.sin(ph)))
    verts.append((0.0, 0.69, 0.0))
    verts.append((0.0, -0.69, 0.0))
    for v in verts:
        lines.append("v %.5f %.5f %.5f" % v)
    top, bottom = len(verts) - 1, len(verts)
    for j in range(nv - 2):
        for i in range(nu):
            a = j * nu + i + 1
            b2 = j * nu + (i + 1) % nu + 1
            lines.append("f %d %d %d %d" % (a, b2, b2 + nu, a + nu))
    for i in range(nu):
        a, b2 = i + 1, (i + 1) % nu + 1
        lines.append("f %d %d %d" % (top, b2, a))
        a2, b3 = (nv - 2) * nu + i + 1, (nv - 2) * nu + (i + 1) % nu + 1
        lines.append("f %d %d %d" % (bottom, a2, b3))
    return "\n".join(lines) + "\n"


def here_be_dragons_objects(api, obj_text=None, nu=48, nv=32, threshold=4):
    """demos/src/bin/here_be_dragons.rs:36-337: six copies of one parsed mesh, each on a pedestal, five of them inside
    a transparent display case that casts no shadow; every element a GroupShape divided with threshold 4."""
    from .obj_parser import parse_obj
    if obj_text is None:
        obj_text = dragon_stand_in_obj(nu, nv)
    pi = PI

    def dragon_material(rgb):
        return api.Material(color=rgb, ambient=0.1, diffuse=0.6, specular=0.3, shininess=15.0)

    def case_material(diffuse, transparency):
        return api.Material(ambient=0.0, diffuse=diffuse, specular=0.0, transparency=transparency)

    elements = [  # (element transform, dragon colour, display case material) -- :41-141
        (api.chain(api.translation(0.0, 0.5, -4.0), api.rotation_y(pi)), (1, 1, 1), None),
        (api.translation(0.0, 2.0, 2.0), (1, 0, 0.1), case_material(0.4, 0.6)),
        (api.chain(api.translation(-2.0, 0.75, -1.0), api.rotation_y(-pi / f32(8.0)), api.scaling(0.75, 0.75, 0.75)),
         (0.9, 0.5, 0.1), case_material(0.2, 0.8)),
        (api.chain(api.translation(-4.0, 0.0, -2.0), api.rotation_y(-pi / f32(16.0)), api.scaling(0.5, 0.5, 0.5)),
         (1, 0.9, 0.1), case_material(0.1, 0.9)),
        (api.chain(api.translation(2.0, 1.0, -1.0), api.rotation_y(f32(5.0) * pi / f32(4.0)), api.scaling(0.75, 0.75, 0.75)),
         (1, 0.5, 0.1), case_material(0.2, 0.8)),
        (api.chain(api.translation(4.0, 0.0, -2.0), api.rotation_y(f32(21.0) * pi / f32(20.0)), api.scaling(0.5, 0.5, 0.5)),
         (0.9, 1, 0.1), case_material(0.1, 0.9)),
    ]
    objects = []
    for transform, rgb, case in elements:
        dragon = parse_obj(obj_text, api).take_all_as_group()          # get_dragon, :290-304 (the demo clones one parse)
        dragon.set_transformation(api.translation(0.0, 0.69, 0.0))
        element = api.GroupShape()                                     # get_scene_element, :306-337
        element.set_transformation(transform)
        dragon.set_material(dragon_material(rgb))
        if case is not None:
            display_case = api.Cube(api.chain(api.scaling(1.1, 0.77, 0.49), api.translation(0.0, 1.001, 0.0)), case,
                                    casts_shadow=False)                # :246-254
            dragon_box = api.GroupShape()
            dragon_box.add_child(dragon)
            dragon_box.add_child(display_case)
        else:
            dragon_box = dragon
        element.add_child(dragon_box)
        element.add_child(api.Cylinder(api.identity_4x4(), api.Material(color=(0.2, 0.2, 0.2), ambient=0.0, diffuse=0.8,
                                                                        specular=0.0, reflective=0.2),
                                       minimum_y=-0.15, maximum_y=0.0, closed=True))  # :269-288
        element.divide(threshold)
        objects.append(element)
    return objects


def here_be_dragons(width=1000, height=400, obj_text=None, nu=48, nv=32):
    """The BVH bonus-chapter demo (here_be_dragons.rs; camera :211-216, light :240-242, depth 5 :218)."""
    from . import api
    world = World(here_be_dragons_objects(api, obj_text, nu, nv), PointLight(point(-10, 100, -100), color(1, 1, 1)))
    camera = Camera(width, height, f32(1.2), view_transform(point(0, 2.5, -10), point(0, 1, 0), vector(0, 1, 0)))
    return world, camera, 5


def synthetic_ppm(width, height, seed=1, scale=255):
    """P3 text of a deterministic test image (smooth bands + blocks + speckle): stands in for the earth / skybox
    photographs the reference's texture demos read from files that are not in its repository."""
    lines = ["P3", "# synthetic test image %dx%d seed %d" % (width, height, seed), "%d %d" % (width, height), str(scale)]
    s = (seed * 2654435761) & 0xFFFFFFFF
    for y in range(height):
        row = []
        for x in range(width):
            s = _xorshift32(s or 1)
            r = (x * scale) // max(width - 1, 1)
            g = (y * scale) // max(height - 1, 1)
            b = ((x // 8 + y // 8) % 2) * (scale // 2) + (s >> 24) * (scale // 2) // 255
            row.append("%d %d %d" % (r, g, b))
        for k in range(0, len(row), 5):
            lines.append("  ".join(row[k:k + 5]))
    return "\n".join(lines) + "\n"


def first_textures_objects(api, earth_ppm=None):
    """World.objects of demos/src/bin/first_textures.rs:37-118; `earth_ppm`: P3 text of the image the demo takes on its
    command line (default: a synthetic 128x64 image)."""
    black, white = (0, 0, 0), (1, 1, 1)
    floor = api.Plane(api.scaling(10.0, 0.01, 10.0),
                      api.Material(specular=0.0, pattern=api.TextureMap(api.UVCheckers(16.0, 8.0, black, white), api.PlanarMap())))
    sphere = api.Sphere(api.translation(-2.5, 1.3, 3.0),
                        api.Material(pattern=api.TextureMap(api.UVCheckers(16.0, 8.0, black, white), api.SphericalMap()),
                                     diffuse=0.7, specular=0.3))
    canvas = api.canvas_from_ppm(earth_ppm if earth_ppm is not None else synthetic_ppm(128, 64))
    earth = api.Sphere(api.chain(api.translation(0.0, 1.0, 0.0), api.rotation_x(f32(-0.5)), api.rotation_y(f32(-1.5))),
                       api.Material(pattern=api.TextureMap(api.UVImage(canvas), api.SphericalMap()), diffuse=0.9, specular=0.1,
                                    shininess=10.0, ambient=0.1))
    pedestal = api.Cylinder(None, api.Material(color=(0.2, 0.2, 0.2), ambient=0.0, diffuse=0.8, specular=0.0, reflective=0.2),
                            minimum_y=-0.15, maximum_y=0.0, closed=True)
    earth_display = api.GroupShape.with_children([earth, pedestal])
    earth_display.set_transformation(api.translation(-0.2, 0.15, 0.5))
    cylinder = api.Cylinder(api.translation(2.0, 2.0, 2.0),
                            api.Material(ambient=0.1, specular=0.6, shininess=15.0, diffuse=0.8,
                                         pattern=api.TextureMap(api.UVCheckers(16.0, 16.0, (0, 0.5, 0), white), api.CylindricalMap())),
                            minimum_y=-3.0, maximum_y=3.0)
    cube = api.Cube(api.chain(api.translation(5.0, 2.0, 2.0), api.rotation_x(-PI / f32(4.0))),
                    api.Material(pattern=align_check_cubic_map(api)))
    return [floor, sphere, cylinder, cube, earth_display]


def align_check_cubic_map(api):
    """get_align_check_cubic_map_pattern, pattern/uv.rs:321-338"""
    white, red, yellow, green, cyan, blue, purple, brown = ((1, 1, 1), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 1, 1), (0, 0, 1),
                                                             (1, 0, 1), (1, 0.5, 0))
    left = api.AlignCheck(yellow, cyan, red, blue, brown)
    front = api.AlignCheck(cyan, red, yellow, brown, green)
    right = api.AlignCheck(red, yellow, purple, green, white)
    back = api.AlignCheck(green, purple, cyan, white, blue)
    up = api.AlignCheck(brown, cyan, purple, red, yellow)
    down = api.AlignCheck(purple, brown, green, blue, white)
    return api.CubicMap(front, back, left, right, up, down)


def first_textures(width=1000, height=500, jitter=("hashed", DEFAULT_SEED), earth_ppm=None):
    """demos/src/bin/first_textures.rs:34-141 (area light :160-170, which the demo draws with thread_rng)."""
    from . import api
    light = RectangleLight(color(1.5, 1.5, 1.5), point(-10, 10, -10), vector(2, 0, 0), 10, vector(0, 2, 0), 10, jitter)
    world = World(first_textures_objects(api, earth_ppm), light)
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0, 1.5, -10), point(2, 2.8, 0), vector(0, 1, 0)))
    return world, camera, 5


def skybox_objects(api, face_size=64):
    """World.objects of demos/src/bin/skybox.rs:40-117 with six synthetic face images."""
    sphere = api.Sphere(api.chain(api.scaling(0.75, 0.75, 0.75), api.translation(0.0, 0.0, 5.0)),
                        api.Material(diffuse=0.4, specular=0.6, shininess=20.0, reflective=0.6, ambient=0.0))
    faces = [api.UVImage(api.canvas_from_ppm(synthetic_ppm(face_size, face_size, seed=k + 2))) for k in range(6)]
    sky = api.Cube(api.scaling(1000.0, 1000.0, 1000.0),
                   api.Material(diffuse=0.0, specular=0.0, ambient=1.0, pattern=api.CubicMap(*faces)))
    return [sphere, sky]


def skybox(width=800, height=400, face_size=64):
    """demos/src/bin/skybox.rs:40-129 (its default canvas is 800x400)."""
    from . import api
    world = World(skybox_objects(api, face_size), PointLight(point(0, 100, 0), color(1, 1, 1)))
    camera = Camera(width, height, 1.2, view_transform(point(0, 0, 0), point(0, 0, 5), vector(0, 1, 0)))
    return world, camera, 5


def shapes_medley(width=256, height=192, jitter=("hashed", 7)):
    """All four shape kinds, nested transparent objects, a non-casting object and an area light:
    a parity stress scene (not a reference demo).  The cylinders are the reflect_refract.rs one
    (get_cylinder, :144-158) plus a closed glass one."""
    floor = Plane(identity_4x4(), Material(color=(0.9, 0.9, 1.0), specular=0.1, reflective=0.25))
    mirror_cyl = Cylinder(chain(translation(2.2, 0.0, 2.0), scaling(0.33, 1.8, 0.33)),
                          Material(reflective=1.0, color=(0.5, 0.5, 0.5), shininess=300.0, specular=0.8),
                          minimum_y=0.0, maximum_y=1.5)
    glass_cyl = Cylinder(chain(translation(-2.0, 0.0, 1.0), scaling(0.6, 1.0, 0.6)),
                         Material(color=(0.05, 0.1, 0.05), transparency=0.9, refractive_index=1.52, reflective=0.3,
                                  diffuse=0.3),
                         minimum_y=0.0, maximum_y=1.2, closed=True)
    outer = Sphere(chain(translation(0.0, 1.0, 0.5), scaling(1.0, 1.0, 1.0)),
                   Material(color=(0, 0, 0.05), specular=1.0, shininess=300.0, transparency=1.0,
                            refractive_index=1.52, reflective=0.9))
    inner = Sphere(chain(translation(0.0, 1.0, 0.5), scaling(0.5, 0.5, 0.5)),
                   Material(color=(0.1, 0, 0), transparency=1.0, refractive_index=1.00029, reflective=0.2,
                            diffuse=0.2))
    box = Cube(chain(translation(-0.6, 0.4, -1.2), rotation_y(f32(0.5)), scaling(0.4, 0.4, 0.4)),
               Material(color=(1.0, 0.6, 0.1), diffuse=0.7, specular=0.3, shininess=50.0))
    ghost = Cube(chain(translation(1.2, 0.5, -1.0), scaling(0.3, 0.5, 0.3)),
                 Material(color=(0.2, 0.9, 0.3), ambient=0.3), casts_shadow=False)
    light = RectangleLight(color(1.2, 1.2, 1.2), point(-3, 5, -4), vector(1.5, 0, 0), 4, vector(0, 0, 1.5), 3, jitter)
    world = World([floor, mirror_cyl, glass_cyl, outer, inner, box, ghost], light)
    camera = Camera(width, height, PI / f32(3.0), view_transform(point(0.5, 2.2, -6.0), point(0, 0.9, 0), vector(0, 1, 0)))
    return world, camera, 5


CONFIGS = {
    "C1": lambda **kw: soft_shadows(1000, 400, **kw),
    "C2": lambda **kw: single_sphere(1024, 1024, **kw),
    "C3": lambda **kw: soft_shadows(4096, 4096, **kw),
    "C4": lambda **kw: glass_and_mirror(4096, 4096, **kw),
    "C5": lambda **kw: sphere_grid(8192, 8192, **kw),
}
