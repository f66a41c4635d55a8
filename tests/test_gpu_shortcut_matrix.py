"""Directed tests of every margin-guarded decision of the render path (ERROR_BUDGET.md), one rule at a time (-m gpu).

Each test builds the smallest scene in which ONE rule decides, puts shade points (or rays) exactly ON that rule's decision
boundary -- a light sample grazing a caster at 1 +- 1e-7 .. 1e-1 of its radius, a shade point 1e-7 .. 0.2 radii off a
sphere, a non-caster just in front of / just behind a caster, a light right at a group's box -- and walks the matrix the
budget is written for: the object's size r (1e-3 .. 1e2), the distance of the shade point in radii (on the surface .. 1e4),
the light's size and distance in radii, and how far the whole scene sits from the world's origin (0 .. 1e4).  Every value
is compared with the oracle's, through the batched entry points rtc_intensity_at / rtc_is_shadowed / rtc_color_at (the
device functions the render kernels inline -- the SIMPLE instantiation included, which is the one that takes the fast
decision of a sample).  A failure names the rule and the cell of the matrix.

What the wide fuzz (tests/test_gpu_fuzz_wide.py) finds by chance, these find by construction."""
import itertools

import numpy as np
import pytest

import ray_tracer_challenge_amd as P
from tests import helpers as H

pytestmark = pytest.mark.gpu
f32 = np.float32

RADII = [1e-3, 0.1, 1.0, 100.0]
OFFSETS = [0.0, 100.0, 1e4]          # |scene centre|; cells where f32 no longer resolves a twentieth of the object are skipped
DIST_IN_RADII = [1.0002, 1.05, 1.5, 4.0, 30.0, 95.0, 105.0, 1e3, 1e4]
LIGHT_DIST_IN_RADII = [1.02, 3.0, 40.0]
LIGHT_SIZE_IN_RADII = [0.02, 1.0, 30.0]
GRAZE = [0.0, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 1e-3, -1e-3, 1e-2, -1e-2, 0.03, -0.03, 0.1, -0.1]


def _resolvable(r, off):
    return off * 2.0 ** -23 <= 0.05 * r


def _unit(rng):
    v = rng.normal(size=3)
    return v / np.linalg.norm(v)


def _perp(rng, a):
    w = np.cross(a, _unit(rng))
    return w / np.linalg.norm(w)


def _mat(**kw):
    return P.Material(color=(0.8, 0.7, 0.6), ambient=0.1, diffuse=0.7, specular=0.0, **kw)


def _area_light(corner, u, v, steps, jitter):
    return P.RectangleLight(P.color(1.0, 1.0, 1.0), P.point(*[float(x) for x in corner]), P.vector(*[float(x) for x in u]), steps[0],
                            P.vector(*[float(x) for x in v]), steps[1], jitter)


def _check_intensity(world, pts, what):
    pts = np.concatenate([np.asarray(pts, dtype=np.float64), np.ones((len(pts), 1))], axis=1).astype(f32)
    got = world.intensity_at(pts)
    own = H.oracle_world(world)
    bad = []
    for i in range(len(pts)):
        own.set_pixel(i)
        exp = own.intensity_at(pts[i])
        if not (got[i] == exp or (np.isnan(got[i]) and np.isnan(exp))):
            bad.append((i, [float(x) for x in pts[i][:3]], float(got[i]), float(exp)))
    assert not bad, "%s: %d of %d light intensities differ, first: point %s gpu %r oracle %r" % (what, len(bad), len(pts), bad[0][1], bad[0][2], bad[0][3])


def _grazing_points(rng, c, r, kind_radius, light_pts, dist, n_each=2):
    """Shade points at `dist` from c from which one of `light_pts` is seen along a line passing c at (1 + g) * kind_radius
    for every g of GRAZE: p = q + (q - L) / |q - L| * sqrt(dist^2 - rho^2), q the tangent point of the sphere of radius rho."""
    out = []
    for g in GRAZE:
        rho = kind_radius * (1.0 + g)
        for _ in range(n_each):
            L = light_pts[int(rng.integers(0, len(light_pts)))]
            a = c - L
            A = np.linalg.norm(a)
            if not (A > rho and dist >= rho):
                continue
            ah = a / A
            w = _perp(rng, ah)
            phi = np.arccos(rho / A)
            q = c + rho * (np.cos(phi) * (-ah) + np.sin(phi) * w)
            along = (q - L) / np.linalg.norm(q - L)
            out.append(q + along * np.sqrt(max(dist * dist - rho * rho, 0.0)))
    return out


@pytest.mark.parametrize("off", OFFSETS)
@pytest.mark.parametrize("r", RADII)
@pytest.mark.parametrize("kind", ["sphere", "ellipsoid", "cube", "cylinder"])
def test_samples_grazing_a_caster(kind, r, off):
    """B1 light-cone cull, B4 fast decision, and the exact path between them: light samples that pass a caster at 1 +- g of its
    size, seen from on its surface to 1e4 radii away, for lights from a speck to thirty radii, touching it or far."""
    if not _resolvable(r, off):
        pytest.skip("f32 does not resolve a twentieth of this object at this offset")
    rng = np.random.default_rng(int(r * 1e4) + int(off) + len(kind))
    c = _unit(rng) * off
    scale = {"sphere": (r, r, r), "ellipsoid": (r, 0.6 * r, 1.4 * r), "cube": (r, r, r), "cylinder": (r, r, r)}[kind]
    t = P.chain(P.translation(*[float(x) for x in c]), P.scaling(*[float(x) for x in scale]))
    ob = {"sphere": P.Sphere, "ellipsoid": P.Sphere, "cube": P.Cube}.get(kind)
    shape = ob(t, _mat()) if ob else P.Cylinder(t, _mat(), minimum_y=-1.0, maximum_y=1.0, closed=True)
    # the radius the grazing lines are laid at: the silhouette of a sphere; for the others what their extremes reach
    rho0 = {"sphere": r, "ellipsoid": 1.4 * r, "cube": r * np.sqrt(3.0), "cylinder": r * np.sqrt(2.0)}[kind]
    for ld, ls, jitter in itertools.product(LIGHT_DIST_IN_RADII, LIGHT_SIZE_IN_RADII, [("hashed", 7), ("constant", 0.5), ("constant", 1.0)]):
        up = _unit(rng)
        u = _perp(rng, up) * ls * r
        v = np.cross(up, u / np.linalg.norm(u)) * ls * r
        corner = c + up * (rho0 * ld + 0.0) - 0.5 * (u + v) if ld > 1.5 else c + up * rho0 * ld
        steps = [(4, 4), (4, 3), (6, 2)][int(rng.integers(0, 3))]  # (even counts both ways: the block cones, B10; otherwise the plain loop)
        world = P.World([shape], _area_light(corner, u / steps[0], v / steps[1], steps, jitter))
        light_pts = [corner + u * a + v * b for a in (0.0, 0.5, 1.0) for b in (0.0, 0.5, 1.0)]
        pts = []
        for dr in DIST_IN_RADII:
            pts += _grazing_points(rng, c, r, rho0 if kind != "sphere" else r, light_pts, dr * rho0)
        _check_intensity(world, pts, "grazing %s r=%g offset=%g light at %g radii, size %g radii, jitter %s" % (kind, r, off, ld, ls, jitter))


@pytest.mark.parametrize("off", OFFSETS)
@pytest.mark.parametrize("r", RADII)
@pytest.mark.parametrize("uniform", [True, False])
def test_shade_points_on_and_just_off_a_sphere(r, off, uniform):
    """B2 `leaving` / B3 `dark` (`entering`): shade points 1e-7 .. 0.25 radii off a casting sphere, all around it -- facing
    the light, on the limb, on the far side -- for lights whose pyramid points away, into, or across the sphere."""
    if not _resolvable(r, off):
        pytest.skip("f32 does not resolve a twentieth of this object at this offset")
    rng = np.random.default_rng(int(r * 1e4) + int(off) + 17 * uniform)
    c = _unit(rng) * off
    s = (r, r, r) if uniform else (r, 0.7 * r, 1.3 * r)
    inv_s = np.array(s)
    sphere = P.Sphere(P.chain(P.translation(*[float(x) for x in c]), P.scaling(*[float(x) for x in s])), _mat())
    for ld, ls, jitter in itertools.product([1.01, 1.3, 2.3, 5.0, 60.0], [0.02, 0.5, 3.0], [("hashed", 3), ("constant", 0.0), ("constant", 1.0)]):
        up = _unit(rng)
        u = _perp(rng, up) * ls * r
        v = np.cross(up, u / np.linalg.norm(u)) * ls * r
        corner = c + up * r * ld - 0.5 * (u + v)
        steps = [(4, 4), (4, 3), (2, 6)][int(rng.integers(0, 3))]
        world = P.World([sphere], _area_light(corner, u / steps[0], v / steps[1], steps, jitter))
        pts = []
        for eps in [1e-7, 1e-6, 1e-5, 9e-5, 1.1e-4, 1e-3, 1.2e-3, 1e-2, 0.09, 0.11, 0.2, 0.25]:
            for _ in range(14):
                d = _unit(rng)
                pts.append(c + d * inv_s * (1.0 + eps))
            for ang in (-0.05, -0.01, 0.0, 0.01, 0.05):  # around the terminator as the light's centre sees it
                w = _perp(rng, up)
                d = np.cos(np.pi / 2 + ang) * up + np.sin(np.pi / 2 + ang) * w
                pts.append(c + d * inv_s * (1.0 + eps))
        _check_intensity(world, pts, "on a sphere r=%g offset=%g uniform=%s light at %g radii, size %g, jitter %s" % (r, off, uniform, ld, ls, jitter))


@pytest.mark.parametrize("off", OFFSETS)
@pytest.mark.parametrize("r", RADII)
def test_a_non_caster_just_in_front_of_or_behind_a_caster(r, off):
    """B5 non-casters behind casters: a lampshade-like non-caster whose nearest point is 1 +- 1e-6 .. 0.3 of the way to the far
    side of a caster, in front of it, around the light; shade points from two to 1e4 radii away."""
    if not _resolvable(r, off):
        pytest.skip("f32 does not resolve a twentieth of this object at this offset")
    rng = np.random.default_rng(int(r * 1e4) + int(off) + 5)
    c = _unit(rng) * off
    caster = P.Sphere(P.chain(P.translation(*[float(x) for x in c]), P.scaling(r, r, r)), _mat())
    for dr, gap, kind in itertools.product([2.0, 10.0, 90.0, 300.0, 1e4], [-0.3, -1e-2, -1e-4, -1e-6, 0.0, 1e-6, 1e-4, 1e-2, 0.3], ["sphere", "slab"]):
        axis = _unit(rng)
        p0 = c - axis * dr * r  # the shade points sit around p0 and look along +axis, past the caster, at the light
        # the non-caster's nearest point from p0 is (1 + gap) x the distance to the caster's far side
        near = (dr + 1.0) * r * (1.0 + gap)
        rn = 0.5 * r
        if kind == "sphere":
            shade = P.Sphere(P.chain(P.translation(*[float(x) for x in (p0 + axis * (near + rn))]), P.scaling(rn, rn, rn)), _mat(), casts_shadow=False)
        else:  # a thin axis-aligned plate, as the demo's lampshade (soft_shadows.rs:97-109); it stands across the world's z
            axis = np.array([0.0, 0.0, 1.0])
            p0 = c - axis * dr * r
            shade = P.Cube(P.chain(P.translation(*[float(x) for x in (p0 + axis * (near + 0.01 * r))]), P.scaling(2 * r, 2 * r, 0.01 * r)), _mat(), casts_shadow=False)
        w = _perp(rng, axis)
        u, v = w * r, np.cross(axis, w) * r
        corner = p0 + axis * (near + 3.0 * r) - 0.5 * (u + v)
        steps = (4, 4) if rng.random() < 0.5 else (3, 3)
        world = P.World([caster, shade], _area_light(corner, u / steps[0], v / steps[1], steps, ("hashed", 11)))
        pts = [p0 + (w * rng.uniform(-1.2, 1.2) + np.cross(axis, w) * rng.uniform(-1.2, 1.2)) * r for _ in range(40)]
        _check_intensity(world, pts, "non-caster (%s) at %+g of the caster's far side, r=%g offset=%g, %g radii away" % (kind, gap, r, off, dr))


@pytest.mark.parametrize("off", OFFSETS)
@pytest.mark.parametrize("scale", [1e-2, 1.0, 1e2])
def test_planes_at_the_lights_own_height(scale, off):
    """B1 (plane rule) and B4 (plane branch of the fast decision): planes just above / below / through an area light's height
    range, shade points just above / below the plane; samples running almost parallel to a plane; the light's samples at
    parameter 1 +- 1e-7 .. 1e-2 of where the ray meets the plane."""
    if not _resolvable(scale, off):
        pytest.skip("f32 does not resolve a twentieth of this scene at this offset")
    rng = np.random.default_rng(int(scale * 100) + int(off))
    c = _unit(rng) * off
    S = scale
    for dy, jitter, tilt in itertools.product([-1e-2, -1e-4, -1e-6, 0.0, 1e-6, 1e-4, 1e-2, 0.5, 1.0, 1.0 + 1e-6, 1.0 - 1e-6, 1.5], [("hashed", 5), ("constant", 0.0), ("constant", 1.0)],
                                              [False, True]):
        # the light spans heights [0, S] above c (a vertical rectangle) or lies flat at height 0
        u = np.array([S, 0.0, 0.0])
        v = np.array([0.0, S, 0.0]) if not tilt else np.array([0.0, 0.0, S])
        corner = c + np.array([-0.5 * S, 0.0, 2.0 * S])
        plane_y = c[1] + dy * S
        floor = P.Plane(P.translation(0.0, float(plane_y), 0.0), _mat())
        ball = P.Sphere(P.chain(P.translation(*[float(x) for x in (c + np.array([0.0, 0.4 * S, 1.0 * S]))]), P.scaling(0.2 * S, 0.2 * S, 0.2 * S)), _mat())
        steps = (4, 4) if rng.random() < 0.5 else (4, 3)
        world = P.World([floor, ball], _area_light(corner, u / steps[0], v / steps[1], steps, jitter))
        pts = []
        for h in [1.2e-3 * S, -1.2e-3 * S, 1e-6 * S, -1e-6 * S, 0.0, 0.3 * S, -0.3 * S, 1e-2 * S]:
            for _ in range(12):
                pts.append(np.array([c[0] + rng.uniform(-3, 3) * S, plane_y + h, c[2] + rng.uniform(-3, 6) * S * 10.0 ** rng.uniform(0, 3)]))
        _check_intensity(world, pts, "plane %+g light heights %s the light, scale %g offset %g jitter %s" % (dy, "under flat" if tilt else "across", scale, off, jitter))


def _check_shadowed(world, lights, points, what):
    lights = np.concatenate([np.asarray(lights, dtype=np.float64), np.ones((len(lights), 1))], axis=1).astype(f32)
    points = np.concatenate([np.asarray(points, dtype=np.float64), np.ones((len(points), 1))], axis=1).astype(f32)
    got = world.is_shadowed(lights, points)
    own = H.oracle_world(world)
    bad = [(i, bool(got[i])) for i in range(len(points)) if bool(got[i]) != bool(own.is_shadowed(lights[i], points[i]))]
    assert not bad, "%s: %d of %d is_shadowed answers differ, first: light %s point %s gpu %r" % (
        what, len(bad), len(points), lights[bad[0][0]][:3], points[bad[0][0]][:3], bad[0][1])


@pytest.mark.parametrize("off", OFFSETS)
@pytest.mark.parametrize("r", RADII)
@pytest.mark.parametrize("divided", [False, True])
def test_a_light_right_at_a_groups_box(r, off, divided):
    """B6 distance pruning: shadow rays whose light sits 1 +- 1e-7 .. 1e-1 of the way to where the ray enters a group's box, or to
    the first hit inside it -- for spheres, cylinders and cones of size r in groups, from one to 1e4 radii away, grazing and
    head-on (a grazing f32 quadratic reports its hit up to 1e-3 of the distance early, ERROR_BUDGET.md E4)."""
    if not _resolvable(r, off):
        pytest.skip("f32 does not resolve a twentieth of this object at this offset")
    rng = np.random.default_rng(int(r * 1e4) + int(off) + 3 * divided)
    c = _unit(rng) * off
    kids = []
    centres = []
    for k in range(5 if divided else 2):
        ck = c + rng.uniform(-3, 3, 3) * r
        centres.append(ck)
        t = P.chain(P.translation(*[float(x) for x in ck]), P.rotation_y(float(rng.uniform(-1, 1))), P.scaling(r, r * float(rng.uniform(0.7, 1.3)), r))
        kind = [P.Sphere, P.Cylinder, P.Cone][k % 3]
        kw = dict(minimum_y=-1.0, maximum_y=1.0, closed=True) if kind is not P.Sphere else {}
        kids.append(kind(t, _mat(), **kw))
    g = P.GroupShape.with_children(kids)
    if divided:
        g.divide(2)
    world = P.World([g], P.PointLight(P.point(0.0, 0.0, 0.0), P.color(1, 1, 1)))
    lights, points = [], []
    for dr in [1.5, 6.0, 50.0, 400.0, 4e3, 1e4]:
        for _ in range(60):
            ck = centres[int(rng.integers(0, len(centres)))]
            dirn = _unit(rng)
            p = ck - dirn * dr * r
            # aim: through the centre, or grazing at 1 +- g radii
            g_off = float(rng.choice(GRAZE)) if rng.random() < 0.7 else -1.0 + rng.uniform(0, 0.9)
            aim = ck + _perp(rng, dirn) * r * (1.0 + g_off)
            ray = (aim - p) / np.linalg.norm(aim - p)
            t_mid = np.linalg.norm(aim - p)
            for rel in [1.0, 1.0 - 1e-7, 1.0 + 1e-7, 1.0 - 1e-5, 1.0 + 1e-5, 1.0 - 1e-3, 1.0 + 1e-3, 0.9, 1.1, 1.0 - r / t_mid, 1.0 + r / t_mid, 1.0 - 3 * r / t_mid,
                        1.0 + 3 * r / t_mid, 2.0]:
                lights.append(p + ray * t_mid * rel)
                points.append(p)
    _check_shadowed(world, lights, points, "light at a group's box, r=%g offset=%g divided=%s" % (r, off, divided))


@pytest.mark.parametrize("off", [0.0, 100.0, 1e4])
@pytest.mark.parametrize("scale", [1e-3, 1.0, 1e3])
def test_rays_grazing_the_triangles_of_a_divided_mesh(scale, off):
    """B7 triangle pre-culling: rays that pass the edges and vertices of a divided mesh's triangles at 0 +- 1e-7 .. 1e-2 of their
    size, head-on and at a few degrees to their planes, from near and from hundreds of mesh sizes away."""
    if not _resolvable(scale, off):
        pytest.skip("f32 does not resolve a twentieth of this mesh at this offset")
    from ray_tracer_challenge_amd import scenes
    from ray_tracer_challenge_amd.obj_parser import parse_obj
    rng = np.random.default_rng(int(scale * 1000) + int(off))
    c = _unit(rng) * off
    g = parse_obj(scenes.bumpy_mesh_obj(7, 5, True), P).take_all_as_group()
    g.set_material(_mat())
    g.set_transformation(P.chain(P.translation(*[float(x) for x in c]), P.rotation_y(0.3), P.scaling(scale, scale, scale)))
    g.divide(3)
    world = P.World([g], P.PointLight(P.point(*[float(x) for x in (c + np.array([3.0, 5.0, -4.0]) * scale)]), P.color(1, 1, 1)))
    own = H.oracle_world(world)
    leaves = g.leaves()
    o, d = [], []
    for _ in range(700):
        tri = leaves[int(rng.integers(0, len(leaves)))]
        fwd = np.asarray(tri.transform, dtype=np.float64)
        pts = [(fwd @ np.append(np.asarray(q, dtype=np.float64)[:3], 1.0))[:3] for q in tri.points]
        a, b = rng.integers(0, 3), rng.integers(0, 3)
        on_edge = pts[a] + (pts[b] - pts[a]) * rng.uniform(0, 1)  # a == b: a vertex
        size = max(np.linalg.norm(pts[1] - pts[0]), np.linalg.norm(pts[2] - pts[0]))
        n = np.cross(pts[1] - pts[0], pts[2] - pts[0])
        n /= max(np.linalg.norm(n), 1e-300)
        target = on_edge + _unit(rng) * size * float(rng.choice([0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2]))
        # direction: anywhere, or within a few degrees of the triangle's plane (where the guard must hand over to the exact test)
        dirn = _unit(rng)
        if rng.random() < 0.4:
            inpl = _perp(rng, n)
            dirn = inpl * np.cos(rng.uniform(0, 0.12)) + n * np.sin(rng.uniform(-0.12, 0.12))
            dirn /= np.linalg.norm(dirn)
        far = scale * 10.0 ** rng.uniform(-1, 2.7)
        o.append(target - dirn * far)
        d.append(dirn)
    o = np.concatenate([np.asarray(o), np.ones((len(o), 1))], axis=1).astype(f32)
    d = np.concatenate([np.asarray(d), np.zeros((len(d), 1))], axis=1).astype(f32)
    got = world.color_at(o, d, 1)
    bad = []
    for i in range(len(o)):
        own.set_pixel(i)
        exp = own.color_at(o[i], d[i], 1)
        if not ((got[i] == exp) | (np.isnan(got[i]) & np.isnan(exp))).all():
            bad.append((i, got[i], exp))
    assert not bad, "mesh scale %g offset %g: %d of %d rays differ, first: origin %s direction %s gpu %s oracle %s" % (
        scale, off, len(bad), len(o), o[bad[0][0]][:3], d[bad[0][0]][:3], bad[0][1], bad[0][2])
