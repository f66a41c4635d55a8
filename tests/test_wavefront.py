"""The level-by-level renderer (csrc/rtc_wavefront.h, -m gpu): one lane per RAY instead of one lane per pixel, a suspended
shade_hit (world.rs:62-86) being a node in HBM and `surface + reflected + refracted` formed bottom-up by a pass per level.
Same rays, same operations per ray and per sum, same jitter keys: images, ray counts and shaded-hit counts must equal the
per-pixel kernels' and the oracle's bit for bit -- whole frames, band partitions (the multi-GPU split), bytes out, every
depth the base stack covers, tree worlds of every kind (GroupShapes, divided meshes, the library's own hierarchy)."""
import os

import numpy as np
import pytest

import ray_tracer_challenge_amd as P
from oracle import oracle as O
from ray_tracer_challenge_amd import scenes
from tests import helpers as H

pytestmark = pytest.mark.gpu
THREADS = min(16, len(os.sched_getaffinity(0)))


def _render(world, camera, depth, mode, monkeypatch, part=None):
    from ray_tracer_challenge_amd.renderer import Renderer
    monkeypatch.setenv("RTC_AMD_WAVEFRONT", mode)
    r = Renderer(world, camera, device=0)
    img = r.render(depth, part=part).cpu().numpy()
    st = r.stats()
    name = r.kernel_name
    r.close()
    return img, st, name


@pytest.mark.parametrize("scene,size", [("mesh", (230, 170)), ("here_be_dragons", (250, 100)), ("hexagons", (200, 100)), ("sphere_grid", (256, 256)),
                                        ("groups_medley", (160, 120)), ("grouped_grid", (192, 192))])
def test_level_by_level_equals_per_pixel_and_oracle(scene, size, monkeypatch):
    world, camera, depth = getattr(scenes, scene)(*size)
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=THREADS)
    by_levels, st1, name1 = _render(world, camera, depth, "1", monkeypatch)
    per_pixel, st0, name0 = _render(world, camera, depth, "0", monkeypatch)
    assert name1.startswith("wavefront[") and not name0.startswith("wavefront["), (name1, name0)
    H.assert_images_equal(by_levels, exp, scene + " level by level")
    H.assert_images_equal(per_pixel, exp, scene + " per pixel")
    assert st1["rays"] == rays == st0["rays"] and st1["shaded_hits"] == st0["shaded_hits"] and st1["pixels"] == st0["pixels"]
    assert st1["culled_shadow_rays"] == st0["culled_shadow_rays"]


@pytest.mark.parametrize("depth", [0, 1, 2, 3, 5, 8])
def test_every_depth_of_the_base_stack(depth, monkeypatch):
    world, camera, _ = scenes.mesh(160, 120)
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=THREADS)
    img, st, name = _render(world, camera, depth, "1", monkeypatch)
    assert name.startswith("wavefront[") == (depth >= 1)  # (depth 0 has no tree to cut: the per-pixel kernel)
    H.assert_images_equal(img, exp, "mesh depth %d" % depth)
    assert st["rays"] == rays


def test_band_partitions_and_bytes(monkeypatch):
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = scenes.mesh(200, 150)
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=THREADS)
    monkeypatch.setenv("RTC_AMD_WAVEFRONT", "1")
    r = Renderer(world, camera, device=0)
    got = np.zeros_like(exp)
    cursor, total = [0, 0, 0], 0
    parts = [r.render(depth, part=Renderer.partition(16, 3, p)).cpu().numpy() for p in range(3)]
    for p in range(3):
        r.render(depth, part=Renderer.partition(16, 3, p))
        total += r.stats()["rays"]
    for b in range((camera.height + 15) // 16):
        p, y0, y1 = b % 3, b * 16, min((b + 1) * 16, camera.height)
        got[y0:y1] = parts[p][cursor[p]:cursor[p] + (y1 - y0)]
        cursor[p] += y1 - y0
    H.assert_images_equal(got, exp, "three band partitions, level by level")
    assert total == rays and r.kernel_name.startswith("wavefront[")
    r.close()
    # through the one-call seam, f32 and bytes (rtc_render_ex falls back to "the stream is done" for launches that cannot report progress)
    canvas = camera.render(world, depth)
    H.assert_images_equal(canvas.data, exp, "one call")
    q = camera.render(world, depth, quantize=True)
    assert np.array_equal(q, O.quantize(exp)) and camera.last_stats["rays"] == rays


def test_it_is_an_alternative_not_the_default(monkeypatch):
    """Measured slower than the per-pixel kernels on every scene (a launch per level, each ending with its own longest wave:
    profiles/r03_wavefront_ab.txt), so nothing picks it unasked; asked for, a frame at a size with thousands of waves per level
    equals the oracle on sampled rows."""
    from ray_tracer_challenge_amd.renderer import Renderer
    monkeypatch.delenv("RTC_AMD_WAVEFRONT", raising=False)
    world, camera, depth = scenes.mesh(1024, 768)
    r = Renderer(world, camera, device=0)
    r.render(depth)
    assert not r.kernel_name.startswith("wavefront["), r.kernel_name
    r.close()
    img, _, name = _render(world, camera, depth, "1", monkeypatch)
    assert name.startswith("wavefront[")
    oc, ow = H.oracle_camera(camera), H.oracle_world(world)
    for y0, y1 in ((0, 3), (300, 312), (420, 428), (764, 768)):
        exp, _ = oc.render(ow, depth, threads=THREADS, rows=(y0, y1))
        H.assert_images_equal(img[y0:y1], exp[y0:y1], "mesh rows %d..%d" % (y0, y1))
