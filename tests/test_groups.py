"""GroupShape / BoundingBox.

CPU: the product API's own tree logic (transform baking, cached bounds, divide) replays the reference's group and
bounding-box unit tests and agrees bit for bit with the oracle's independent implementation on whole scenes.
GPU (-m gpu): worlds with groups rendered by the packet-traversal kernel equal the oracle's recursive traversal.
"""
import numpy as np
import pytest

import ray_tracer_challenge_amd as P
from oracle import oracle as O
from ray_tracer_challenge_amd import _lib as L
from ray_tracer_challenge_amd import scenes
from tests import helpers as H
from tests import kat as K

f32 = np.float32


def _box(pair):
    return P.BoundingBox(K.point(pair[0]), K.point(pair[1]))


def test_bounds_helpers_known_answers(kat):  # bounding_box.rs:147-288 through rtc_bounds_*
    B = kat["bounding_box"]
    c = B["add_box"]
    b = _box(c["box1"])
    b.add_bounding_box(_box(c["box2"]))
    K.assert_exact(b.min, K.point(c["min"]))
    K.assert_exact(b.max, K.point(c["max"]))
    c = B["contains_box"]
    for mn, mx, expect in c["cases"]:
        assert _box(c["box"]).contains_bounding_box(_box([mn, mx])) == expect
    c = B["contains_point"]  # a degenerate box is a point
    for p, expect in c["cases"]:
        assert _box(c["box"]).contains_bounding_box(_box([p, p])) == expect
    c = B["transform"]
    b2 = _box(c["box"]).transform(P.mat_mul(P.rotation_x(K.CONSTS["PI"] / f32(4.0)), P.rotation_y(K.CONSTS["PI"] / f32(4.0))))
    K.assert_eps(b2.min[:3], c["min_eps"])
    K.assert_eps(b2.max[:3], c["max_eps"])
    for c in B["split"]["cases"]:
        left, right = _box(c["box"]).split()
        for got, want in ((left.min, c["left"][0]), (left.max, c["left"][1]), (right.min, c["right"][0]), (right.max, c["right"][1])):
            K.assert_exact(got, K.point(want))
    c = B["shape_in_parent_space"]
    b = P.Sphere(P.mat_mul(P.translation(1.0, -3.0, 5.0), P.scaling(0.5, 2.0, 4.0))).parent_space_bounding_box()
    K.assert_exact(b.min, K.point(c["min"]))
    K.assert_exact(b.max, K.point(c["max"]))
    b = P.Cone().bounding_box()
    K.assert_exact(b.min, K.point(B["cone_unbounded"]["min"]))
    K.assert_exact(b.max, K.point(B["cone_unbounded"]["max"]))
    c = B["cone_bounded"]
    b = P.Cone(minimum_y=c["min_y"], maximum_y=c["max_y"]).bounding_box()
    K.assert_exact(b.min, K.point(c["min"]))
    K.assert_exact(b.max, K.point(c["max"]))


def test_bounds_match_oracle_on_random_boxes_and_transforms():
    rng = np.random.default_rng(8)
    for _ in range(200):
        lo = rng.uniform(-5, 5, 3).astype(f32)
        hi = lo + rng.uniform(0, 6, 3).astype(f32)
        m = P.chain(P.translation(*rng.uniform(-3, 3, 3)), P.rotation_y(f32(rng.uniform(-3, 3))), P.rotation_x(f32(rng.uniform(-3, 3))),
                    P.scaling(*rng.uniform(0.2, 3, 3)))
        pb, ob = P.BoundingBox(K.point(lo), K.point(hi)), O.BoundingBox(K.point(lo), K.point(hi))
        pt, ot = pb.transform(m), ob.transform(m)
        assert np.array_equal(pt.min, ot.min) and np.array_equal(pt.max, ot.max)
        (pl, pr), (ol, or_) = pb.split(), ob.split()
        for a, b in ((pl, ol), (pr, or_)):
            assert np.array_equal(a.min, b.min) and np.array_equal(a.max, b.max)
    # infinite bounds through a rotation: 0 * inf = NaN corners, ignored by f32::min / max
    for kind_p, kind_o in ((P.Plane, O.Plane), (P.Cylinder, O.Cylinder), (P.Cone, O.Cone)):
        t = P.chain(P.translation(1.0, 2.0, 3.0), P.rotation_z(f32(0.3)))
        a, b = kind_p(t).parent_space_bounding_box(), kind_o(t).parent_space_bounding_box()
        assert np.array_equal(a.min, b.min, equal_nan=True) and np.array_equal(a.max, b.max, equal_nan=True)


def test_group_transform_baking_known_answers(kat):  # shape/group.rs:216-340,607-639
    G = kat["group"]
    g = P.GroupShape.with_children([P.Sphere(), P.Sphere(), P.Sphere()])
    g.set_material(P.Material(shininess=123.456))
    assert [c.material.shininess for c in g.get_children()] == [123.456] * 3
    expect = K.mat(G["baked_child_transform"]["expect_exact"])
    g = P.GroupShape()
    g.set_transformation(P.scaling(2.0, 2.0, 2.0))
    g.add_child(P.Sphere(P.translation(5.0, 0.0, 0.0)))
    K.assert_exact(g.get_children()[0].transformation(), expect)
    g = P.GroupShape()
    g.add_child(P.Sphere(P.translation(5.0, 0.0, 0.0)))
    g.set_transformation(P.scaling(2.0, 2.0, 2.0))
    K.assert_exact(g.get_children()[0].transformation(), expect)
    g = P.GroupShape()
    g.set_transformation(P.scaling(3.0, 4.0, 8.0))
    g.add_child(P.Sphere(P.translation(5.0, 0.0, 0.0)))
    g.set_transformation(P.scaling(2.0, 2.0, 2.0))
    K.assert_exact(g.get_children()[0].transformation(), expect)
    c = G["bounding_box_contains_children"]
    g = P.GroupShape()
    g.add_child(P.Sphere(P.mat_mul(P.translation(2.0, 5.0, -3.0), P.scaling(2.0, 2.0, 2.0))))
    g.add_child(P.Cylinder(P.mat_mul(P.translation(-4.0, -1.0, 4.0), P.scaling(0.5, 1.0, 0.5)), minimum_y=-2.0, maximum_y=2.0))
    K.assert_exact(g.bounding_box().min, K.point(c["min"]))
    K.assert_exact(g.bounding_box().max, K.point(c["max"]))
    c = G["divide_preserves_transformation"]
    group = P.GroupShape()
    group.set_transformation(P.translation(1.0, 1.0, 0.0))
    for t in ((-2.0, 0.0, 0.0), (2.0, -1.0, 0.0), (2.0, 1.0, 0.0)):
        group.add_child(P.Sphere(P.translation(*t)))
    group.divide(2)
    ch = group.get_children()
    K.assert_exact(ch[0].transformation(), P.translation(*c["s1"]))
    K.assert_exact(ch[1].get_children()[0].transformation(), P.translation(*c["s2"]))
    K.assert_exact(ch[1].get_children()[1].transformation(), P.translation(*c["s3"]))


def test_group_divide_structure():  # shape/group.rs:458-604
    s1, s2, s3 = P.Sphere(P.translation(-2.0, -2.0, 0.0)), P.Sphere(P.translation(-2.0, 2.0, 0.0)), P.Sphere(P.scaling(4.0, 4.0, 4.0))
    g = P.GroupShape()
    for s in (s1, s2, s3):
        g.add_child(s)
    g.divide(1)
    ch = g.get_children()
    assert ch[0] is s3 and isinstance(ch[1], P.GroupShape) and ch[1].get_children() == [s1, s2]
    s1, s2, s3 = P.Sphere(P.translation(-2.0, 0.0, 0.0)), P.Sphere(P.translation(2.0, 0.0, 0.0)), P.Sphere()
    g = P.GroupShape()
    for s in (s1, s2, s3):
        g.add_child(s)
    left, right = g._partition_children()
    assert g.get_children() == [s3] and left == [s1] and right == [s2]
    g = P.GroupShape()
    a, b = P.Sphere(), P.Sphere()
    g._make_subgroup([a, b])
    assert len(g.get_children()) == 1 and g.get_children()[0].get_children() == [a, b]
    s1, s2, s3, s4 = (P.Sphere(P.translation(-2.0, 0.0, 0.0)), P.Sphere(P.translation(2.0, 1.0, 0.0)),
                      P.Sphere(P.translation(2.0, -1.0, 0.0)), P.Sphere())
    sub = P.GroupShape()
    for s in (s1, s2, s3):
        sub.add_child(s)
    g = P.GroupShape()
    g.add_child(sub)
    g.add_child(s4)
    g.divide(3)
    ch = g.get_children()
    assert ch[0] is sub and ch[1] is s4
    assert sub.get_children()[0] is s1 and sub.get_children()[1].get_children() == [s2, s3]


def _tree_signature_p(node):
    if isinstance(node, P.GroupShape):
        b = node.bounding_box()
        return ("g", b.min[:3].tobytes(), b.max[:3].tobytes(), [_tree_signature_p(c) for c in node.get_children()])
    return ("s", node.kind, np.asarray(node.transformation(), dtype=f32).tobytes())


def _tree_signature_o(node):
    if node.is_group:
        b = node.bounding_box()
        return ("g", b.min[:3].tobytes(), b.max[:3].tobytes(), [_tree_signature_o(c) for c in node.get_children()])
    return ("s", None, node.transformation().tobytes())


def _strip_kind(sig):
    return ("g", sig[1], sig[2], [_strip_kind(c) for c in sig[3]]) if sig[0] == "g" else ("s", None, sig[2])


@pytest.mark.parametrize("name", ["hexagons", "grouped_grid", "groups_medley", "here_be_dragons"])
def test_product_and_oracle_build_identical_trees(name):
    """The same construction script run against the product API and against the oracle's API: same tree shape,
    bit-identical baked leaf transforms and group bounding boxes (two independent implementations of
    add_child / set_transformation / divide / bounding_box)."""
    build = getattr(scenes, name + "_objects")
    kw = {"nu": 12, "nv": 8} if name == "here_be_dragons" else {}
    for p_node, o_node in zip(build(P, **kw), build(O, **kw)):
        if isinstance(p_node, P.GroupShape):
            assert _strip_kind(_tree_signature_p(p_node)) == _tree_signature_o(o_node)


def test_flattened_groups_layout():
    world, _, _ = scenes.hexagons(32, 16)
    cs = world._c()
    assert cs.scene.n_objects == 13 and cs.scene.n_groups == 7
    runs = [(cs.groups[i].first_object, cs.groups[i].n_objects) for i in range(7)]
    assert runs == [(1, 12)] + [(1 + 2 * k, 2) for k in range(6)]
    outer = world.objects[1].bounding_box()
    assert list(cs.groups[0].bounds_min) == list(outer.min[:3]) and list(cs.groups[0].bounds_max) == list(outer.max[:3])


# ------------------------------------------------------------------------------------------- GPU
def _oracle_light(lt):
    if hasattr(lt, "corner"):
        return O.RectangleLight(lt.intensity, lt.corner, lt.u_vec, lt.u_steps, lt.v_vec, lt.v_steps, lt.jitter)
    return O.PointLight(lt.position, lt.intensity)


@pytest.mark.gpu
@pytest.mark.parametrize("name,size,kw", [
    ("hexagons", (200, 100), {}),
    ("hexagons", (61, 47), {}),
    ("grouped_grid", (128, 128), {}),
    ("grouped_grid", (96, 96), {"threshold": 1}),
    ("groups_medley", (128, 96), {}),
    ("groups_medley", (64, 48), {"jitter": ("constant", 0.5)}),
])
def test_group_scenes_match_oracle_bitwise(name, size, kw):
    world, camera, depth = getattr(scenes, name)(*size, **kw)
    canvas = camera.render(world, depth)
    stats = camera.last_stats
    oc = H.oracle_camera(camera)
    img, rays = oc.render(H.oracle_world(world), depth, threads=8)
    H.assert_images_equal(canvas.data, img, name)
    assert stats["rays"] == rays
    assert canvas.to_ppm() == O.to_ppm(img)
    # and against a world the ORACLE built itself from the same script (its own baking / divide / bounds)
    okw = {k: v for k, v in kw.items() if k != "jitter"}
    own = O.World(getattr(scenes, name + "_objects")(O, **okw), _oracle_light(world.light))
    img2, rays2 = oc.render(own, depth, threads=8)
    H.assert_images_equal(canvas.data, img2, name + " (oracle-built tree)")
    assert rays2 == rays


@pytest.mark.gpu
def test_group_kernel_selection_and_flat_equivalence():
    """A world whose only group holds everything and whose box every ray hits... is still gated: compare the tree
    kernel with the flat kernels where gating cannot matter (box = whole scene, camera inside it)."""
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = scenes.hexagons(64, 32)
    r = Renderer(world, camera, device=0)
    assert r.kernel_name == "render_kernel<tree>"
    flat_world, camera2, _ = scenes.first_scene(64, 32)
    assert Renderer(flat_world, camera2, device=0).kernel_name.startswith("render_kernel<")
    # empty groups are ignored: a world of one empty group plus flat shapes renders with the flat kernels
    w = P.World([P.GroupShape()] + list(flat_world.objects), flat_world.light)
    r2 = Renderer(w, camera2, device=0)
    assert r2.kernel_name != "render_kernel<tree>"
    a = r2.render(depth).cpu().numpy()
    b = Renderer(flat_world, camera2, device=0).render(depth).cpu().numpy()
    assert np.array_equal(a, b)


@pytest.mark.gpu
def test_batched_entry_points_walk_the_tree():
    """rtc_color_at / rtc_is_shadowed / rtc_intensity_at on a world with groups against the oracle."""
    world, camera, depth = scenes.groups_medley(32, 24, jitter=("constant", 0.5))
    ow = H.oracle_world(world)
    rng = np.random.default_rng(4)
    n = 300
    o = np.concatenate([rng.uniform(-4, 4, (n, 2)), rng.uniform(-8, -5, (n, 1)), np.ones((n, 1))], axis=1).astype(f32)
    d = np.concatenate([rng.uniform(-0.5, 0.5, (n, 2)), np.ones((n, 1)), np.zeros((n, 1))], axis=1).astype(f32)
    d = np.array([O.norm(v) for v in d], dtype=f32)
    got = world.color_at(o, d, depth)
    for i in range(n):
        ow.set_pixel(i)
        exp = ow.color_at(o[i], d[i], depth)
        assert np.array_equal(got[i], exp), (i, got[i], exp)
    pts = np.concatenate([rng.uniform(-3, 3, (n, 1)), rng.uniform(0.01, 2, (n, 1)), rng.uniform(-3, 3, (n, 1)), np.ones((n, 1))], axis=1).astype(f32)
    lp = np.tile(np.array([-4.0, 6.0, -5.0, 1.0], dtype=f32), (n, 1))
    sh = world.is_shadowed(lp, pts)
    assert [bool(x) for x in sh] == [ow.is_shadowed(lp[i], pts[i]) for i in range(n)]


@pytest.mark.gpu
def test_malformed_groups_are_refused():
    world, camera, depth = scenes.hexagons(16, 8)
    cs = world._c()
    cam = camera._cam
    out = np.zeros((8, 16, 3), dtype=f32)

    def render():
        return L.lib().rtc_render(cs.scene, cam, depth, 0, out.ctypes.data_as(L.FP), None)
    import ctypes as C
    L.lib().rtc_render.argtypes = [C.POINTER(L.rtc_scene), C.POINTER(L.rtc_camera), C.c_int32, C.c_int32, L.FP, C.c_void_p]
    assert render() == L.RTC_OK
    cs.groups[1].n_objects = 13          # side 0 sticks out of the hexagon
    assert render() == L.RTC_ERR_INVALID_ARG
    cs.groups[1].n_objects = 2
    cs.groups[0].first_object = 2        # not pre-order any more
    assert render() == L.RTC_ERR_INVALID_ARG
    cs.groups[0].first_object = 1
    cs.groups[6].first_object = 40       # outside the object list
    assert render() == L.RTC_ERR_INVALID_ARG


def _small_tree_world(seed, area_light):
    """<= 8 leaves under <= 8 (nested) groups, some leaves outside any group: the case the unrolled kernels take with
    the groups' boxes as gates."""
    rng = np.random.default_rng(seed)

    def leaf():
        kind = rng.choice(["sphere", "cube", "cylinder", "sphere"])
        t = P.chain(P.translation(*[float(v) for v in rng.uniform(-2.0, 2.0, 3)]), P.rotation_y(float(rng.uniform(-3, 3))),
                    P.scaling(*[float(v) for v in rng.uniform(0.3, 0.9, 3)]))
        u = rng.random()
        m = P.Material(color=tuple(rng.uniform(0.2, 1.0, 3)), reflective=0.4 if u < 0.3 else 0.0,
                       transparency=0.7 if 0.3 <= u < 0.55 else 0.0, refractive_index=1.4)
        if kind == "cylinder":
            return P.Cylinder(t, m, minimum_y=-1.0, maximum_y=1.0, closed=bool(rng.random() < 0.5), casts_shadow=bool(rng.random() < 0.8))
        return (P.Sphere if kind == "sphere" else P.Cube)(t, m, casts_shadow=bool(rng.random() < 0.8))

    def group(depth, budget):
        g = P.GroupShape()
        g.set_transformation(P.chain(P.translation(*[float(v) for v in rng.uniform(-0.5, 0.5, 3)]), P.rotation_z(float(rng.uniform(-0.5, 0.5)))))
        n = int(rng.integers(1, 4))
        for _ in range(n):
            if budget[0] <= 0:
                break
            if depth < 2 and rng.random() < 0.35 and budget[1] > 0:
                budget[1] -= 1
                g.add_child(group(depth + 1, budget))
            else:
                budget[0] -= 1
                g.add_child(leaf())
        return g

    budget = [int(rng.integers(3, 8)), 5]  # leaves, groups
    objs = [P.Plane(P.translation(0.0, -2.5, 0.0), P.Material(color=(0.8, 0.8, 0.8), reflective=0.2))]
    while budget[0] > 0:
        if rng.random() < 0.7 and budget[1] > 0:
            budget[1] -= 1
            objs.append(group(0, budget))
        else:
            budget[0] -= 1
            objs.append(leaf())
    if area_light:
        light = P.RectangleLight(P.color(1, 1, 1), P.point(-2, 5, -4), P.vector(2, 0, 0), 3, P.vector(0, 1, 1), 3, ("hashed", seed))
    else:
        light = P.PointLight(P.point(-4, 7, -6), P.color(1, 1, 1))
    camera = P.Camera(120, 90, float(np.pi / 3), P.view_transform(P.point(0.5, 2.0, -7.5), P.point(0, 0, 0), P.vector(0, 1, 0)))
    return P.World(objs, light), camera


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("specialise", ["0", "1"])
def test_small_trees_through_the_unrolled_kernels(seed, specialise, monkeypatch):
    """Group boxes as gates (rtc_device.hip flatten / SceneHdr::gate_mask) against the oracle's recursive walk and
    against the traversal kernel (RTC_AMD_GATES=0)."""
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera = _small_tree_world(seed, area_light=seed % 2 == 0)
    n_leaves = len(world._c().leaves)
    assert n_leaves <= 8
    monkeypatch.setenv("RTC_AMD_SPECIALIZE", specialise)
    out = {}
    for gates in ("1", "0"):
        monkeypatch.setenv("RTC_AMD_GATES", gates)
        r = Renderer(world, camera, device=0)
        has_groups = any(isinstance(o, P.GroupShape) and o.leaves() for o in world.objects)
        if has_groups:
            assert ("tree" in r.kernel_name) == (gates == "0"), r.kernel_name
        out[gates] = (r.render(4).cpu().numpy(), r.stats())
        r.close()
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), 4, threads=8)
    for gates in ("1", "0"):
        H.assert_images_equal(out[gates][0], exp, "seed %d gates=%s" % (seed, gates))
        assert out[gates][1]["rays"] == rays
    assert out["1"][1]["shaded_hits"] == out["0"][1]["shaded_hits"]


def _stale_box_objects(api, many):
    """group.rs:15,138-151: a group's bounding box is cached on first use and never invalidated.  A child added AFTER
    that lies outside the box the group keeps testing rays against: it is visible only to rays that also pass
    through the stale box.  The one place where a group's box visibly decides what a ray sees."""
    g = api.GroupShape()
    g.add_child(api.Sphere(api.translation(-1.5, 0.0, 0.0), api.Material(color=(1, 0.2, 0.2))))
    g.bounding_box()  # cached here: around the first sphere only
    g.add_child(api.Sphere(api.chain(api.translation(1.2, 0.4, 0.0), api.scaling(1.3, 1.3, 1.3)), api.Material(color=(0.2, 1, 0.2))))
    g.add_child(api.Cube(api.chain(api.translation(-1.5, 1.8, 0.5), api.scaling(0.5, 0.5, 0.5)), api.Material(color=(0.2, 0.2, 1), reflective=0.3)))
    objs = [g, api.Plane(api.translation(0.0, -1.0, 0.0), api.Material(color=(0.9, 0.9, 0.9), reflective=0.3))]
    if many:  # more than 8 leaves: the traversal kernel
        for k in range(7):
            objs.append(api.Sphere(api.chain(api.translation(-4.0 + k * 1.3, -0.6, 3.0), api.scaling(0.4, 0.4, 0.4)), api.Material()))
    return objs


@pytest.mark.gpu
@pytest.mark.parametrize("many", [False, True])
def test_a_stale_cached_box_hides_late_children_like_the_reference(many, monkeypatch):
    from ray_tracer_challenge_amd.renderer import Renderer
    light = ((-5.0, 8.0, -6.0), (1.0, 1.0, 1.0))
    camera = P.Camera(160, 120, float(np.pi / 3), P.view_transform(P.point(0, 1.0, -7), P.point(0, 0.3, 0), P.vector(0, 1, 0)))
    world = P.World(_stale_box_objects(P, many), P.PointLight(P.point(*light[0]), P.color(*light[1])))
    own = O.World(_stale_box_objects(O, many), O.PointLight(np.array(light[0] + (1.0,), dtype=f32), np.array(light[1], dtype=f32)))
    exp, rays = H.oracle_camera(camera).render(own, 3, threads=8)
    fresh = P.World(_stale_box_objects(P, many), world.light)
    fresh.objects[0]._cached_box = None  # what the scene would look like if the box were recomputed
    modes = ("1", "0") if not many else ("1",)
    for gates in modes:
        monkeypatch.setenv("RTC_AMD_GATES", gates)
        r = Renderer(world, camera, device=0)
        assert ("tree" in r.kernel_name) == (many or gates == "0"), r.kernel_name
        img = r.render(3).cpu().numpy()
        H.assert_images_equal(img, exp, "stale box, gates=%s many=%s" % (gates, many))
        assert r.stats()["rays"] == rays
        r.close()
        seen = Renderer(fresh, camera, device=0).render(3).cpu().numpy()
        assert (seen != img).any(axis=2).mean() > 0.02  # the late children really are hidden from most rays


@pytest.mark.gpu
def test_shortcuts_that_assert_a_hit_respect_a_stale_group_box():
    """'Every sample is blocked by the sphere this point sits behind' (light_cull_mask's far-side shortcut) presumes the
    rays reach the sphere.  A sphere added to a group after the group's box was cached lies outside that box, and the
    reference turns its rays away at the group: points on its far side are LIT there, and must be here."""
    light = ((-3.0, 6.0, -5.0), (1.0, 1.0, 1.0))

    def build(api):
        objs = _stale_box_objects(api, False)
        lt = api.RectangleLight(np.array(light[1], dtype=f32), np.array(light[0] + (1.0,), dtype=f32), np.array((2, 0, 0, 0), dtype=f32), 4,
                                np.array((0, 2, 0, 0), dtype=f32), 4, ("hashed", 4))
        return api.World(objs, lt)

    world, own = build(P), build(O)
    late = world.objects[0].children[1]  # the sphere added after bounding_box() was cached
    rng = np.random.default_rng(3)
    d = rng.normal(size=(3000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    local = np.concatenate([d * (1.0 + 10.0 ** rng.uniform(-5, -1, (3000, 1))), np.ones((3000, 1))], axis=1)
    pts = (np.asarray(late.transform, dtype=np.float64) @ local.T).T.astype(f32)
    got = world.intensity_at(pts)
    for i in range(len(pts)):
        own.set_pixel(i)
        exp = own.intensity_at(pts[i])
        assert got[i] == exp, (pts[i], got[i], exp)
    assert (got == 1.0).mean() > 0.5  # most points around the hidden sphere are lit, its own far side included
