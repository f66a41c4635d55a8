"""Random worlds far outside the demos' comfort zone -- the generator behind tests/test_gpu_fuzz_wide.py and
tools/fuzz_wide.py.

tests/test_gpu_fuzz.py draws translations in +-3 and scales in 0.25 .. 1.2: the regime the reference's demos live in, and
one in which every conservative shortcut of the render path (ERROR_BUDGET.md: light-cone culling and its `dark` / `away` /
`leaving` rules, the fast shadow decision, non-casters behind casters, distance pruning of groups, triangle pre-culling,
the scene box, the library's own hierarchy) has margins to spare.  The one parity break the project has had (a sphere of
0.25 seen from 4 000 units, commit 133fd44) was outside it.  Here, per style:

  0 cluster   a handful of flat objects, log-uniform sizes 1e-3 .. 1e3, the whole scene up to 1e4 from the world origin
  1 thin      the same with one axis scaled by 1e-3 .. 1e-1 (the demo's own lampshade is scaling(1, 1, 0.01),
              soft_shadows.rs:97-109): plates, discs, needles, non-casters around the light
  2 horizon   a floor seen from just above it: shade points from one to 1e5 scene units out, small casters at the light
  3 touching  lights that almost touch a caster (gap 1e-4 .. 1e-1 of its radius), shade points right under the light
  4 groups    nested / divided groups whose own transforms scale by 1e-2 .. 1e2
  5 mesh      a divided mesh scaled 1e-3 .. 1e3 and moved up to 1e4 away
  6 many      16 .. 40 flat objects seen from 5 .. 300 scene sizes away through a narrow lens (hierarchy, scene box)
  7 grazing   a narrow lens from far away aimed at a silhouette: every primary ray nearly tangent to a sphere, nearly
              parallel to a cube face, a cylinder wall, a plane or a triangle

`world(seed, api)` builds the same description against the product's API and against the oracle's.  Magnitudes stay where
f32 squares of object-space coordinates do not overflow (<= 1e7 object units): a world point at +-inf is outside the
arithmetic contract.
"""
import numpy as np

from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.obj_parser import parse_obj

N_STYLES = 8
STYLE_NAMES = ["cluster", "thin", "horizon", "touching", "groups", "mesh", "many", "grazing"]


def _lu(rng, lo, hi):
    return float(10.0 ** rng.uniform(np.log10(lo), np.log10(hi)))


def _unit(rng):
    v = rng.normal(size=3)
    return v / np.linalg.norm(v)


def _centre(rng, S):
    """Where the scene sits: at the world origin, or up to 1e4 away from it -- but not so far that f32 coordinates no longer
    resolve a hundredth of the scene (a frame of one flat colour compares equal and tests nothing)."""
    far = min(1e4, 3e4 * S)
    if rng.random() < 0.3 or far <= 1.0:
        return np.zeros(3)
    return _unit(rng) * _lu(rng, 1.0, far)


def _material(api, rng, glassy=0.15):
    u = rng.random()
    return api.Material(color=tuple(float(v) for v in rng.uniform(0.1, 1.0, 3)), ambient=float(rng.uniform(0.05, 0.3)),
                        diffuse=float(rng.uniform(0.4, 0.9)), specular=float(rng.choice([0.0, 0.0, 0.3, 0.9])),
                        shininess=float(rng.choice([10.0, 50.0, 200.0])), reflective=float(rng.uniform(0.2, 0.9)) if u < 0.2 else 0.0,
                        transparency=float(rng.uniform(0.4, 0.95)) if 0.2 <= u < 0.2 + glassy else 0.0,
                        refractive_index=float(rng.choice([1.0, 1.33, 1.5, 2.4])))


def _place(api, rng, at, size, rotate, thin=False, uniform=None):
    """translation(at) [* rotations] * scaling(size ...)"""
    s = np.full(3, size)
    if uniform is None:
        uniform = rng.random() < 0.5
    if not uniform:
        s = s * rng.uniform(0.5, 1.5, 3)
    if thin:
        s[int(rng.integers(0, 3))] *= _lu(rng, 1e-3, 1e-1)
        if rng.random() < 0.3:
            s[int(rng.integers(0, 3))] *= _lu(rng, 1e-2, 1e-1)  # a needle
    ms = [api.translation(*[float(v) for v in at])]
    if rotate:
        ms += [api.rotation_y(float(rng.uniform(-3, 3))), api.rotation_x(float(rng.uniform(-1.5, 1.5)))]
        if rng.random() < 0.3:
            ms.append(api.rotation_z(float(rng.uniform(-1.5, 1.5))))
    ms.append(api.scaling(*[float(v) for v in s]))
    return api.chain(*ms)


def _leaf(api, rng, at, size, rotate, thin=False, kinds=("sphere", "sphere", "cube", "cylinder"), casts=None, uniform=None, glassy=0.15):
    kind = str(rng.choice(list(kinds)))
    t = _place(api, rng, at, size, rotate, thin, uniform)
    m = _material(api, rng, glassy)
    casts = bool(rng.random() < 0.85) if casts is None else casts
    if kind == "sphere":
        return api.Sphere(t, m, casts_shadow=casts)
    if kind == "cube":
        return api.Cube(t, m, casts_shadow=casts)
    if kind == "plane":
        return api.Plane(t, m, casts_shadow=casts)
    if kind == "triangle":
        pts = [api.point(*[float(v) for v in rng.uniform(-1.5, 1.5, 3)]) for _ in range(3)]
        return api.Triangle(*pts, t, m, casts_shadow=casts)
    lo = float(rng.uniform(-1.5, 0.0))
    kw = dict(minimum_y=lo, maximum_y=lo + float(rng.uniform(0.3, 2.0)), closed=bool(rng.random() < 0.6)) if rng.random() < 0.9 else {}
    return (api.Cylinder if kind == "cylinder" else api.Cone)(t, m, casts_shadow=casts, **kw)


def _light(api, rng, seed, at, size, area=None, steps=(2, 7), aligned=None):
    """A point light at `at`, or an area light of edge ~`size` whose corner is `at`."""
    if area is None:
        area = rng.random() < 0.65
    if not area:
        return api.PointLight(api.point(*[float(v) for v in at]), api.color(1.0, 1.0, 1.0))
    jitter = ("hashed", seed) if rng.random() < 0.6 else ("constant", float(rng.choice([0.0, 0.5, 1.0])))
    if aligned is None:
        aligned = rng.random() < 0.5
    if aligned:  # axis-aligned, as the demo's: the kernels compiled for it drop the zero components (LIGHT_ZEROS)
        axes = rng.permutation(3)
        u, v = np.zeros(3), np.zeros(3)
        u[axes[0]] = size * rng.uniform(0.5, 1.5) * rng.choice([-1.0, 1.0])
        v[axes[1]] = size * rng.uniform(0.5, 1.5) * rng.choice([-1.0, 1.0])
    else:
        u = _unit(rng)
        v = np.cross(u, _unit(rng))
        u, v = u * size * rng.uniform(0.5, 1.5), v / np.linalg.norm(v) * size * rng.uniform(0.5, 1.5)
    return api.RectangleLight(api.color(1.1, 1.0, 0.9), api.point(*[float(x) for x in at]), api.vector(*[float(x) for x in u]),
                              int(rng.integers(*steps)), api.vector(*[float(x) for x in v]), int(rng.integers(*steps)), jitter)


def _camera(api, rng, frm, to, fov=None, size=None):
    w, h = size if size is not None else (int(rng.integers(40, 80)), int(rng.integers(30, 60)))
    fov = float(rng.uniform(0.6, 1.3)) if fov is None else float(fov)
    up = api.vector(0.0, 1.0, 0.0) if rng.random() < 0.8 else api.vector(*[float(v) for v in _unit(rng)])
    return (w, h, fov, api.view_transform(api.point(*[float(v) for v in frm]), api.point(*[float(v) for v in to]), up))


def _floor(api, rng, C, S, tilt=True):
    t = [api.translation(float(C[0]), float(C[1] - rng.uniform(1.0, 2.5) * S), float(C[2]))]
    if tilt and rng.random() < 0.4:
        t.append(api.rotation_z(float(rng.uniform(-0.2, 0.2))))
    return api.Plane(api.chain(*t), _material(api, rng, glassy=0.0))


def world(seed, api):
    """-> (World, camera tuple (w, h, fov, transform), depth, style name)"""
    rng = np.random.default_rng(77000 + seed)
    style = seed % N_STYLES
    S = _lu(rng, 1e-2, 1e2)  # the scene's own size
    C = _centre(rng, S)
    rotate = rng.random() < 0.45  # otherwise every object is scale + translate only: the SIMPLE kernels, shadow_fast
    objs = []
    depth = int(rng.integers(0, 4))
    light = camera = None

    def size_of():  # mostly within two decades of the scene, sometimes anything in 1e-3 .. 1e3
        return S * _lu(rng, 0.03, 1.5) if rng.random() < 0.75 else _lu(rng, 1e-3, 1e3)

    if style in (0, 1):
        if rng.random() < 0.7:
            objs.append(_floor(api, rng, C, S))
        n = int(rng.integers(1, 7 - len(objs)))
        for k in range(n):
            objs.append(_leaf(api, rng, C + rng.uniform(-3, 3, 3) * S, size_of(), rotate, thin=(style == 1 and rng.random() < 0.7)))
        lat = C + (rng.uniform(-3, 3, 3) + np.array([0.0, 5.0, -2.0])) * S
        light = _light(api, rng, seed, lat, S * _lu(rng, 0.05, 3.0))
        if style == 1 and rng.random() < 0.6:  # a lampshade: a thin non-caster right behind / around an area light
            objs.append(_leaf(api, rng, lat + rng.uniform(-0.5, 0.5, 3) * S, S * rng.uniform(0.5, 2.0), rotate, thin=True, kinds=("cube",), casts=False))
        camera = _camera(api, rng, C + (rng.uniform(-2, 2, 3) + np.array([0.0, 1.5, -9.0])) * S, C)
    elif style == 2:
        floor = api.Plane(api.chain(api.translation(*[float(v) for v in C]), api.rotation_z(float(rng.uniform(-0.01, 0.01)))),
                          _material(api, rng, glassy=0.0))
        objs.append(floor)
        lat = C + np.array([rng.uniform(-1, 1), rng.uniform(1.5, 4.0), rng.uniform(2.0, 6.0)]) * S
        for k in range(int(rng.integers(1, 5))):  # small casters at the light
            objs.append(_leaf(api, rng, lat + rng.uniform(-1.5, 1.5, 3) * S * np.array([1.0, 0.5, 1.0]), S * _lu(rng, 0.01, 0.5), rotate,
                              kinds=("sphere", "sphere", "cylinder", "cube")))
        light = _light(api, rng, seed, lat, S * _lu(rng, 0.05, 2.0))
        eye_h = S * _lu(rng, 0.01, 1.0)
        frm = C + np.array([rng.uniform(-1, 1) * S, eye_h, -6.0 * S])
        to = C + np.array([rng.uniform(-1, 1) * S, eye_h * rng.uniform(0.5, 1.0), 6.0 * S * _lu(rng, 1.0, 1e3)])
        camera = _camera(api, rng, frm, to, fov=_lu(rng, 0.02, 0.8), size=(int(rng.integers(60, 100)), int(rng.integers(24, 40))))
    elif style == 3:
        objs.append(_floor(api, rng, C, S, tilt=False))
        r = S * _lu(rng, 0.05, 1.0)
        at = C + rng.uniform(-1, 1, 3) * S
        kind = str(rng.choice(["sphere", "sphere", "cube", "cylinder"]))
        objs.append(_leaf(api, rng, at, r, rotate, kinds=(kind,), casts=True, uniform=True))
        gap = r * _lu(rng, 1e-4, 1e-1)
        side = _unit(rng)
        side[1] = abs(side[1])  # above the caster, mostly
        lsize = r * _lu(rng, 0.01, 2.0)
        lat = at + side * (r * (1.0 if kind == "sphere" else 1.75) + gap)
        light = _light(api, rng, seed, lat, lsize)
        for k in range(int(rng.integers(0, 3))):
            objs.append(_leaf(api, rng, C + rng.uniform(-3, 3, 3) * S, size_of(), rotate))
        camera = _camera(api, rng, C + (rng.uniform(-2, 2, 3) + np.array([0.0, 2.5, -8.0])) * S, at)
    elif style == 4:
        if rng.random() < 0.6:
            objs.append(_floor(api, rng, C, S))
        budget = [int(rng.integers(4, 16))]

        def group(level, scale):
            g = api.GroupShape()
            gs = _lu(rng, 1e-2, 1e2) if rng.random() < 0.5 else 1.0
            g.set_transformation(_place(api, rng, (C if level == 0 else np.zeros(3)) + rng.uniform(-2, 2, 3) * scale, gs, rng.random() < 0.5, uniform=True))
            inner = scale / gs  # the children's coordinates are in the group's space
            for _ in range(int(rng.integers(1, 6))):
                if budget[0] <= 0:
                    break
                if level < 2 and rng.random() < 0.3:
                    g.add_child(group(level + 1, inner))
                else:
                    budget[0] -= 1
                    g.add_child(_leaf(api, rng, rng.uniform(-2, 2, 3) * inner, inner * _lu(rng, 0.03, 1.0), rng.random() < 0.5,
                                      kinds=("sphere", "sphere", "cube", "cylinder", "cone", "triangle")))
            # (dividing a subgroup BEFORE its parent adopts it caches a box that the re-baking leaves stale, group.rs:15 -- the
            # reference then turns rays away from where the children are: kept as a rare case, it leaves little to see)
            if level > 0 and rng.random() < 0.15:
                g.divide(int(rng.integers(1, 4)))
            return g

        while budget[0] > 0:
            g = group(0, S)
            if rng.random() < 0.7:
                g.divide(int(rng.integers(1, 4)))
            objs.append(g)
        light = _light(api, rng, seed, C + (rng.uniform(-3, 3, 3) + np.array([0.0, 6.0, -3.0])) * S, S * _lu(rng, 0.05, 2.0), steps=(2, 7))
        camera = _camera(api, rng, C + (rng.uniform(-2, 2, 3) + np.array([0.0, 1.5, -10.0])) * S, C)
    elif style == 5:
        for k in range(int(rng.integers(1, 3))):
            text = scenes.bumpy_mesh_obj(int(rng.integers(5, 9)), int(rng.integers(4, 7)), bool(rng.random() < 0.5))
            g = parse_obj(text, api).take_all_as_group()
            g.set_material(_material(api, rng, glassy=0.3))
            g.set_transformation(_place(api, rng, C + rng.uniform(-1.5, 1.5, 3) * S, S * rng.uniform(0.5, 1.2), True, thin=rng.random() < 0.2, uniform=rng.random() < 0.7))
            g.divide(int(rng.integers(2, 7)))
            objs.append(g)
        if rng.random() < 0.5:
            objs.append(_floor(api, rng, C, S))
        light = _light(api, rng, seed, C + (rng.uniform(-3, 3, 3) + np.array([0.0, 6.0, -3.0])) * S, S * _lu(rng, 0.05, 2.0), area=rng.random() < 0.3, steps=(2, 7))
        far = _lu(rng, 6.0, 300.0)
        camera = _camera(api, rng, C + _unit(rng) * np.array([1.0, 0.5, 1.0]) * far * S, C + rng.uniform(-0.5, 0.5, 3) * S, fov=min(1.2, 4.0 / far))
        depth = int(rng.integers(0, 6))
    elif style == 6:
        for k in range(int(rng.integers(16, 40))):
            objs.append(_leaf(api, rng, C + rng.uniform(-4, 4, 3) * S, S * _lu(rng, 0.01, 1.0) if rng.random() < 0.85 else size_of(), rotate,
                              kinds=("sphere", "sphere", "cube"), uniform=rng.random() < 0.7))
        light = _light(api, rng, seed, C + (rng.uniform(-3, 3, 3) + np.array([0.0, 7.0, -4.0])) * S, S * _lu(rng, 0.05, 2.0), area=rng.random() < 0.3, steps=(2, 7))
        far = _lu(rng, 5.0, 300.0)
        camera = _camera(api, rng, C + _unit(rng) * far * S, C + rng.uniform(-1, 1, 3) * S, fov=min(1.3, 10.0 / far))
    else:  # grazing
        kind = str(rng.choice(["sphere", "sphere", "cube", "cylinder", "plane", "triangle", "cone"]))
        r = S
        at = C.copy()
        ob = _leaf(api, rng, at, r, rotate, kinds=(kind,), casts=True, uniform=True, glassy=0.3)
        objs.append(ob)
        if kind != "plane" and rng.random() < 0.6:
            objs.append(_floor(api, rng, C, 2.0 * S))
        for k in range(int(rng.integers(0, 3))):
            objs.append(_leaf(api, rng, C + rng.uniform(-3, 3, 3) * S, size_of(), rotate))
        far = r * _lu(rng, 10.0, 1e4)
        view = _unit(rng)
        view[1] = abs(view[1]) * 0.3 + 0.02
        view /= np.linalg.norm(view)
        frm = at + view * far
        # a point of the silhouette as seen from `frm`: on the unit shape's rim, perpendicular to the line of sight
        side = np.cross(view, _unit(rng))
        side /= np.linalg.norm(side)
        rim = at + side * r * (1.0 if kind in ("sphere", "cylinder", "cone") else rng.uniform(0.9, 1.5))
        if kind == "plane":
            rim = at + np.array([view[0], 0.0, view[2]]) * (-far * _lu(rng, 1.0, 100.0))  # towards the horizon
        window = r * _lu(rng, 1e-4, 0.3)  # what the image spans at the object
        camera = _camera(api, rng, frm, rim, fov=max(2.0 * np.arctan(0.5 * window / far), 1e-4))  # (pixels stay apart in f32 directions)
        light = _light(api, rng, seed, C + (rng.uniform(-3, 3, 3) + np.array([0.0, 5.0, 0.0])) * S, S * _lu(rng, 0.05, 2.0), steps=(2, 7))
    return api.World(objs, light), camera, depth, STYLE_NAMES[style]
