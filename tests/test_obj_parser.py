"""obj_parser.rs:303-533 replayed against ray_tracer_challenge_amd.obj_parser (host-side input format of the mesh
path), and -- on the GPU -- a parsed mesh rendered through groups of triangles against the oracle."""
import os

import numpy as np
import pytest

import ray_tracer_challenge_amd as P
from oracle import oracle as O
from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.obj_parser import ParseError, parse_obj
from tests import helpers as H
from tests import kat as K

f32 = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))


def _text(block):
    return "\n".join("        " + ln for ln in block["lines"])  # the reference's literals are indented


def _tri_points(t):
    return [np.asarray(p, dtype=f32) for p in t.points]


def test_obj_parser_known_answers(kat):
    B = kat["obj_parser"]
    assert parse_obj(_text(B["gibberish"])).num_ignored_lines == B["gibberish"]["ignored"]
    for key in ("vertex_records", "normalized"):
        r = parse_obj(_text(B[key]))
        assert len(r.vertices) == 5
        for got, want in zip(r.vertices[1:], B[key]["vertices"]):
            K.assert_exact(got, K.point(want))
    for key in ("triangle_faces", "polygon"):
        r = parse_obj(_text(B[key]))
        kids = r.get_default_group().get_children()
        assert len(kids) == len(B[key]["triangles"])
        for t, idx in zip(kids, B[key]["triangles"]):
            for got, i in zip(_tri_points(t), idx):
                K.assert_exact(got, r.vertices[i])
    c = B["groups_file"]
    text = open(os.path.join(HERE, "golden", "triangles.obj")).read()
    r = parse_obj(text)
    for name in ("FirstGroup", "SecondGroup"):
        t = r.get_group(name).get_children()[0]
        for got, i in zip(_tri_points(t), c[name]):
            K.assert_exact(got, r.vertices[i])
    parent = parse_obj(text).take_all_as_group()
    g1, g2 = parent.get_children()
    K.assert_exact(_tri_points(g1.get_children()[0])[0], K.point(c["first_p1"]))
    K.assert_exact(_tri_points(g2.get_children()[0])[0], K.point(c["first_p1"]))
    c = B["one_group"]
    for header in ([], ["g TestGroup"]):
        r = parse_obj("\n".join([""] + c["vertices"] + [""] + header + c["faces"] + [""]))
        assert len(r.take_all_as_group().get_children()) == c["children"]
    r = parse_obj(_text(B["normal_records"]))
    assert len(r.normals) == 4
    for got, want in zip(r.normals[1:], B["normal_records"]["normals"]):
        K.assert_exact(got, K.vector(want))
    r = parse_obj(_text(B["faces_with_normals"]))
    kids = r.get_default_group().get_children()
    assert len(kids) == 2
    for t in kids:
        for got, i in zip(_tri_points(t), (1, 2, 3)):
            K.assert_exact(got, r.vertices[i])
        for got, i in zip(t.normals, (1, 2, 3)):
            K.assert_exact(got, r.normals[i])


def test_obj_parser_errors():
    for text, kind in (("v 1 2", "MalformedVertex"), ("vn 1 2 3 4", "MalformedNormal"), ("v 1 2 3\nv 0 1 0\nv 1 0 0\nf 1 2", "MalformedFace"),
                       ("g", "MalformedGroupDeclaration"), ("v 1 2 3\nv 0 1 0\nv 1 0 0\nf 1 2 3\nv 4 5 6", "UnexpectedSymbol"),
                       ("v a b c", "ParseFloatError"), ("v 1 2 3\nf x 2 3", "ParseIntError"), ("v 1 2 3\nf /1 2 3", "MalformedFace")):
        with pytest.raises(ParseError) as e:
            parse_obj(text)
        assert e.value.kind == kind, text


def test_triangle_fields_and_bounds_known_answers(kat):  # shape/triangle.rs:101-176 through the C ABI helpers
    T = kat["triangle"]
    d = T["default"]
    t = P.Triangle(K.point(d["p1"]), K.point(d["p2"]), K.point(d["p3"]))
    e1, e2, normal = P.triangle_fields(t)
    K.assert_exact(e1, K.vector(T["construction"]["e1"]))
    K.assert_exact(e2, K.vector(T["construction"]["e2"]))
    K.assert_exact(normal, K.vector(T["construction"]["normal"]))
    c = T["bounding_box"]
    b = P.Triangle(*[K.point(p) for p in c["points"]]).bounding_box()
    K.assert_exact(b.min, K.point(c["min"]))
    K.assert_exact(b.max, K.point(c["max"]))
    s = T["smooth"]
    st = P.SmoothTriangle(K.point(d["p1"]), K.point(d["p2"]), K.point(d["p3"]), K.vector(s["n1"]), K.vector(s["n2"]), K.vector(s["n3"]))
    n = st.local_norm_at_uv(s["u_exact"], s["v_exact"])
    K.assert_eps(O.norm(n), s["interpolated_normal_eps"])  # identity transform: normal_to_world only normalises
    rng = np.random.default_rng(21)
    for _ in range(100):  # random triangles under random transforms: fields and parent-space boxes vs the oracle
        pts = [np.append(rng.uniform(-3, 3, 3), 1.0).astype(f32) for _ in range(3)]
        m = P.chain(P.translation(*rng.uniform(-2, 2, 3)), P.rotation_x(f32(rng.uniform(-3, 3))), P.scaling(*rng.uniform(0.3, 2, 3)))
        pt, ot = P.Triangle(*pts, transform=m), O.Triangle(*pts, transform=m)
        for a, b in zip(P.triangle_fields(pt), ot.triangle_fields()):
            assert np.array_equal(a, b)
        pb, ob = pt.parent_space_bounding_box(), ot.parent_space_bounding_box()
        assert np.array_equal(pb.min, ob.min) and np.array_equal(pb.max, ob.max)


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_triangle_known_answers_on_device(kat):  # shape/triangle.rs:113-165 through rtc_local_intersect / rtc_normal_at
    T = kat["triangle"]
    d = T["default"]
    t = P.Triangle(K.point(d["p1"]), K.point(d["p2"]), K.point(d["p3"]))
    got = t.local_intersect([K.point(o) for o, _ in T["misses"]["rays"]], [K.vector(v) for _, v in T["misses"]["rays"]])
    assert got == [[] for _ in T["misses"]["rays"]]
    (ts,) = t.local_intersect([K.point(T["strikes"]["ray"][0])], [K.vector(T["strikes"]["ray"][1])])
    assert ts == [f32(T["strikes"]["distance_exact"])]
    normals = t.normal_at([K.point(p) for p in T["normal_points"]["points"]])
    for n in normals:
        K.assert_exact(n, K.vector(T["construction"]["normal"]))


@pytest.mark.gpu
def test_triangle_intersections_match_oracle_bitwise():
    rng = np.random.default_rng(77)
    for trial in range(6):
        pts = [np.append(rng.uniform(-2, 2, 3), 1.0).astype(f32) for _ in range(3)]
        m = P.chain(P.translation(*rng.uniform(-1, 1, 3)), P.rotation_z(f32(rng.uniform(-3, 3))), P.scaling(*rng.uniform(0.4, 2, 3)))
        pt, ot = P.Triangle(*pts, transform=m), O.Triangle(*pts, transform=m)
        n = 3000
        o = np.concatenate([rng.uniform(-3, 3, (n, 3)), np.ones((n, 1))], axis=1).astype(f32)
        target = np.array([(pts[0] + pts[1] + pts[2])[:3] / 3.0], dtype=f32) + rng.normal(0, 1.2, (n, 3)).astype(f32)
        d = np.concatenate([target - o[:, :3], np.zeros((n, 1))], axis=1).astype(f32)
        d = np.array([O.norm(v) for v in d], dtype=f32)
        got = pt.local_intersect(o, d)
        hits = 0
        for i in range(n):
            exp = ot.local_intersect(o[i], d[i])
            assert len(got[i]) == len(exp) and np.array_equal(np.array(got[i], dtype=f32), np.array(exp, dtype=f32)), (trial, i, got[i], exp)
            hits += len(exp)
        assert hits > n // 20
        wp = np.concatenate([rng.uniform(-2, 2, (50, 3)), np.ones((50, 1))], axis=1).astype(f32)
        gn = pt.normal_at(wp)
        for i in range(50):
            assert np.array_equal(gn[i], ot.normal_at(wp[i]))


@pytest.mark.gpu
@pytest.mark.parametrize("size,kw", [((160, 120), {}), ((96, 72), {"nu": 8, "nv": 6, "threshold": 2}), ((64, 48), {"threshold": 1000})])
def test_mesh_scene_matches_oracle_bitwise(size, kw):
    """Two parsed OBJ meshes (Triangle and SmoothTriangle leaves in divided groups), a loose triangle and a plane."""
    world, camera, depth = scenes.mesh(*size, **kw)
    canvas = camera.render(world, depth)
    oc = H.oracle_camera(camera)
    img, rays = oc.render(H.oracle_world(world), depth, threads=8)
    H.assert_images_equal(canvas.data, img, "mesh")
    assert camera.last_stats["rays"] == rays
    own = O.World(scenes.mesh_objects(O, **kw), O.PointLight(world.light.position, world.light.intensity))
    img2, rays2 = oc.render(own, depth, threads=8)
    H.assert_images_equal(canvas.data, img2, "mesh (oracle-parsed, oracle-built tree)")
    assert rays2 == rays


@pytest.mark.gpu
@pytest.mark.parametrize("size,kw", [((125, 50), {"nu": 12, "nv": 8}), ((250, 100), {"nu": 20, "nv": 12})])
def test_here_be_dragons_scene_matches_oracle_bitwise(size, kw):
    """demos/src/bin/here_be_dragons.rs with a procedural stand-in for its dragon.obj: six divided mesh groups on
    pedestals, five inside transparent display cases that cast no shadow."""
    world, camera, depth = scenes.here_be_dragons(*size, **kw)
    canvas = camera.render(world, depth)
    oc = H.oracle_camera(camera)
    img, rays = oc.render(H.oracle_world(world), depth, threads=8)
    H.assert_images_equal(canvas.data, img, "here_be_dragons")
    assert camera.last_stats["rays"] == rays
    assert canvas.to_ppm() == O.to_ppm(img)
    own = O.World(scenes.here_be_dragons_objects(O, **kw), O.PointLight(world.light.position, world.light.intensity))
    img2, rays2 = oc.render(own, depth, threads=8)
    H.assert_images_equal(canvas.data, img2, "here_be_dragons (oracle-parsed, oracle-built tree)")
    assert rays2 == rays


@pytest.mark.gpu
def test_flat_world_of_triangles_uses_the_unrolled_kernels():
    """Triangles outside any group take the flat kernels (AOT and scene-specialised): a tetrahedron."""
    from ray_tracer_challenge_amd.renderer import Renderer
    a, b, c, d = P.point(0, 1.5, 0), P.point(-1, 0, -1), P.point(1, 0, -1), P.point(0, 0, 1)
    m = P.Material(color=(0.8, 0.3, 0.3), reflective=0.2)
    tris = [P.Triangle(a, b, c, None, m), P.Triangle(a, c, d, None, m), P.Triangle(a, d, b, None, m), P.Triangle(b, d, c, None, m)]
    floor = P.Plane(P.translation(0.0, -0.01, 0.0), P.Material(color=(0.9, 0.9, 0.9), specular=0.0))
    world = P.World(tris + [floor], P.PointLight(P.point(-5, 6, -6), P.color(1, 1, 1)))
    camera = P.Camera(96, 64, scenes.PI / f32(3.0), P.view_transform(P.point(0, 1.5, -5), P.point(0, 0.5, 0), P.vector(0, 1, 0)))
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), 5, threads=8)
    import os as _os
    for mode in ("0", "1"):
        _os.environ["RTC_AMD_SPECIALIZE"] = mode
        try:
            r = Renderer(world, camera, device=0)
            assert r.kernel_name.startswith("render_kernel_spec[" if mode == "1" else "render_kernel<8,general>")
            img = r.render(5).cpu().numpy()
            H.assert_images_equal(img, exp, "tetrahedron specialise=" + mode)
            assert r.stats()["rays"] == rays
        finally:
            del _os.environ["RTC_AMD_SPECIALIZE"]
