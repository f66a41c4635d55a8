"""GPU tests at BASELINE.json's full sizes (-m gpu).

Every pixel of every BASELINE configuration is compared with the CPU oracle, bit for bit, together with the frame's ray
count (the reference's semantics are per pixel, camera.rs:80-85): the row-parallel oracle renders the 4096^2 soft_shadows
frame (1.18 G rays) in about ten seconds on the GPU box's 16 host threads, the other configurations in less.  On top of
that, size-independent properties: partition invariance (any band split assembles to the identical image and the
identical ray count), the never-traced last row / column, determinism, the device-side quantiser against the oracle's
scale_color, and -- for the one shortcut that answers most of the frame's shadow rays -- whole-frame equality of the
image and of every counter with light-cone culling switched off (RTC_AMD_LIGHT_CULL=0) and with only its hit-asserting
`dark` rule switched off (RTC_AMD_DARK=0).
"""
import os

import numpy as np
import pytest

import ray_tracer_challenge_amd as P
from oracle import oracle as O
from ray_tracer_challenge_amd import scenes
from tests import helpers as H

pytestmark = pytest.mark.gpu
f32 = np.float32
THREADS = min(16, len(os.sched_getaffinity(0)))


@pytest.fixture(scope="module")
def torch():
    import torch as t
    return t


def _renderer(world, camera):
    from ray_tracer_challenge_amd.renderer import Renderer
    return Renderer(world, camera, device=0)


def _check_whole_frame(image, rays, world, camera, depth, what):
    """Every pixel of the frame and the frame's ray count against the oracle (row-parallel on the host's cores)."""
    import time
    t0 = time.perf_counter()
    exp, exp_rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=THREADS)
    dt = time.perf_counter() - t0
    print("\n%s: oracle rendered %dx%d (%d rays) in %.1f s on %d threads" % (what, camera.width, camera.height, exp_rays, dt, THREADS))
    if not np.array_equal(image, exp):  # (NaN-free frames: asserted by the callers) -> the detailed report
        H.assert_images_equal(image, exp, what)
    assert rays == exp_rays, (what, rays, exp_rays)


@pytest.fixture
def cull_env():
    """Sets / restores the light-cone culling switches (read by the library when a scene is set)."""
    saved = {k: os.environ.get(k) for k in ("RTC_AMD_LIGHT_CULL", "RTC_AMD_DARK", "RTC_AMD_FAST_SHADOW", "RTC_AMD_CELL_CULL")}

    def set_(**kw):
        for k, v in kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    yield set_
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _assemble(parts, height, n_parts, band_rows=64):
    """Re-interleave compact per-part band buffers into the full image (host side, numpy)."""
    out = np.zeros((height,) + parts[0].shape[1:], dtype=parts[0].dtype)
    cursor = [0] * n_parts
    n_bands = (height + band_rows - 1) // band_rows
    for b in range(n_bands):
        p = b % n_parts
        y0, y1 = b * band_rows, min((b + 1) * band_rows, height)
        out[y0:y1] = parts[p][cursor[p]:cursor[p] + (y1 - y0)]
        cursor[p] += y1 - y0
    return out


def test_c3_soft_shadows_4096(torch):
    world, camera, depth = scenes.CONFIGS["C3"]()
    r = _renderer(world, camera)
    img_t = r.render(depth)
    st = r.stats()
    img = img_t.cpu().numpy()
    assert st["pixels"] == 4095 * 4095 and st["rays"] > 100 * st["shaded_hits"] > 0
    assert not img[-1].any() and not img[:, -1].any()          # camera.rs:80-81
    assert np.isfinite(img).all()
    _check_whole_frame(img, st["rays"], world, camera, depth, "C3 4096^2 hashed jitter")
    # determinism
    again = r.render(depth).cpu().numpy()
    assert np.array_equal(img, again) and r.stats()["rays"] == st["rays"]
    # partition invariance: 2, 3 and 8 parts (the multi-GPU split) give the same image and the same ray count
    for n in (2, 3, 8):
        parts, rays, pixels = [], 0, 0
        for p in range(n):
            part = r.partition(64, n, p)
            parts.append(r.render(depth, part=part).cpu().numpy())
            s = r.stats()
            rays += s["rays"]
            pixels += s["pixels"]
        assert rays == st["rays"] and pixels == st["pixels"], (n, rays, st["rays"])
        assert np.array_equal(_assemble(parts, camera.height, n), img), n
    # device-side quantiser == canvas.rs scale_color, on every channel of the frame
    q = r.quantize(img_t).cpu().numpy()
    assert np.array_equal(q, O.quantize(img))


def test_c3_constant_jitter_whole_frame(torch):
    world, camera, depth = scenes.soft_shadows(4096, 4096, jitter=("constant", 0.5))
    r = _renderer(world, camera)
    img = r.render(depth).cpu().numpy()
    assert np.isfinite(img).all()
    _check_whole_frame(img, r.stats()["rays"], world, camera, depth, "C3 4096^2 constant jitter 0.5")


@pytest.mark.parametrize("config", ["C3 hashed", "C3 constant 0.5", "C1 hashed", "C1 constant 0.5"])
def test_light_cone_cull_changes_nothing_whole_frame(torch, cull_env, config):
    """The cull answers ~79 % of C3's shadow rays without an object test, and the margin-guarded fast decision (shadow_fast)
    most of the rest without normalising the ray (world.rs:104-119, rectangle_light.rs:76-88 are what both must
    preserve).  Whole frames with everything on, everything off, and each shortcut off alone: same image, same rays,
    same shaded hits -- and with the cull off no ray may be counted as culled."""
    size = (4096, 4096) if config.startswith("C3") else (1000, 400)
    jitter = ("hashed", scenes.DEFAULT_SEED) if config.endswith("hashed") else ("constant", 0.5)
    world, camera, depth = scenes.soft_shadows(*size, jitter=jitter)
    frames = {}
    for name, env in (("on", dict(RTC_AMD_LIGHT_CULL=None, RTC_AMD_DARK=None, RTC_AMD_FAST_SHADOW=None, RTC_AMD_CELL_CULL=None)),
                      ("off", dict(RTC_AMD_LIGHT_CULL="0", RTC_AMD_DARK=None, RTC_AMD_FAST_SHADOW="0", RTC_AMD_CELL_CULL=None)),
                      ("no_dark", dict(RTC_AMD_LIGHT_CULL=None, RTC_AMD_DARK="0", RTC_AMD_FAST_SHADOW=None, RTC_AMD_CELL_CULL=None)),
                      ("no_fast", dict(RTC_AMD_LIGHT_CULL=None, RTC_AMD_DARK=None, RTC_AMD_FAST_SHADOW="0", RTC_AMD_CELL_CULL=None)),
                      ("fast_only", dict(RTC_AMD_LIGHT_CULL="0", RTC_AMD_DARK=None, RTC_AMD_FAST_SHADOW=None, RTC_AMD_CELL_CULL=None)),
                      ("no_cells", dict(RTC_AMD_LIGHT_CULL=None, RTC_AMD_DARK=None, RTC_AMD_FAST_SHADOW=None, RTC_AMD_CELL_CULL="0")),
                      ("cells_exact", dict(RTC_AMD_LIGHT_CULL=None, RTC_AMD_DARK="0", RTC_AMD_FAST_SHADOW="0", RTC_AMD_CELL_CULL=None))):
        cull_env(**env)
        r = _renderer(world, camera)
        img = r.render(depth).cpu().numpy()
        frames[name] = (img, r.stats(), r.kernel_name)
        r.close()
    on, off, no_dark, no_fast, fast_only, no_cells, cells_exact = (frames[k] for k in ("on", "off", "no_dark", "no_fast", "fast_only", "no_cells", "cells_exact"))
    assert on[2] == off[2] == no_dark[2] == no_fast[2] == no_cells[2]  # the same kernel: the switches are run-time data
    assert no_cells[1]["culled_shadow_rays"] <= on[1]["culled_shadow_rays"]  # (the cells decide what the whole-light cull left)
    for other, what in ((off, "every shortcut off"), (no_dark, "dark off"), (no_fast, "fast sample decision off"),
                        (fast_only, "cull off, fast sample decision on"), (no_cells, "cell cones off"),
                        (cells_exact, "cell cones with exact samples")):
        assert np.array_equal(on[0], other[0]), (config, what, int((on[0] != other[0]).sum()))
        for key in ("rays", "shaded_hits", "pixels"):
            assert on[1][key] == other[1][key], (config, what, key)
    assert off[1]["culled_shadow_rays"] == 0 and on[1]["culled_shadow_rays"] > 0
    assert no_dark[1]["culled_shadow_rays"] <= on[1]["culled_shadow_rays"]
    print("\n%s: kernel ms all shortcuts %.3f, cell cones off %.3f, dark off %.3f, fast sample decision off %.3f, cull off (fast on) %.3f, all off %.3f; "
          "%d of %d rays answered without a sample (%d without the cell cones)" % (
              config, on[1]["kernel_ms"], no_cells[1]["kernel_ms"], no_dark[1]["kernel_ms"], no_fast[1]["kernel_ms"],
              fast_only[1]["kernel_ms"], off[1]["kernel_ms"], on[1]["culled_shadow_rays"], on[1]["rays"], no_cells[1]["culled_shadow_rays"]))


def test_c2_single_sphere_1024_full_image(torch):
    world, camera, depth = scenes.CONFIGS["C2"]()
    canvas = camera.render(world, depth)
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=THREADS)
    H.assert_images_equal(canvas.data, exp, "C2")
    assert camera.last_stats["rays"] == rays and camera.last_stats["pixels"] == 1023 * 1023
    assert canvas.to_ppm() == O.to_ppm(exp)  # the wire format, byte for byte


def test_c4_glass_and_mirror_4096(torch):
    world, camera, depth = scenes.CONFIGS["C4"]()
    r = _renderer(world, camera)
    img = r.render(depth).cpu().numpy()
    st = r.stats()
    assert st["pixels"] == 4095 * 4095 and np.isfinite(img).all()
    _check_whole_frame(img, st["rays"], world, camera, depth, "C4 4096^2")
    small_w, small_c, _ = scenes.glass_and_mirror(384, 384)
    canvas = small_c.render(small_w, depth)
    exp, rays = H.oracle_camera(small_c).render(H.oracle_world(small_w), depth, threads=THREADS)
    H.assert_images_equal(canvas.data, exp, "C4 384^2")
    assert small_c.last_stats["rays"] == rays


def test_c5_sphere_grid_8192(torch):
    world, camera, depth = scenes.CONFIGS["C5"]()
    r = _renderer(world, camera)
    whole = r.render(depth).cpu().numpy()
    st = r.stats()
    assert st["pixels"] == 8191 * 8191
    assert np.isfinite(whole).all()
    _check_whole_frame(whole, st["rays"], world, camera, depth, "C5 8192^2")
    # the 8-GPU split of BASELINE config 5, rendered part by part on one GPU
    parts, rays = [], 0
    for p in range(8):
        parts.append(r.render(depth, part=r.partition(64, 8, p)).cpu().numpy())
        rays += r.stats()["rays"]
    assert rays == st["rays"]
    assert np.array_equal(_assemble(parts, camera.height, 8), whole)
    del parts, whole
    small_w, small_c, _ = scenes.sphere_grid(512, 512)
    canvas = small_c.render(small_w, depth)
    exp, rays = H.oracle_camera(small_c).render(H.oracle_world(small_w), depth, threads=THREADS)
    H.assert_images_equal(canvas.data, exp, "C5 512^2")
    assert small_c.last_stats["rays"] == rays


def test_c1_soft_shadows_demo_default_resolution(torch):
    """BASELINE config 1 (the demo binary's 1000x400): the whole frame, and the P3 text byte for byte."""
    world, camera, depth = scenes.CONFIGS["C1"]()
    canvas = camera.render(world, depth)
    _check_whole_frame(canvas.data, camera.last_stats["rays"], world, camera, depth, "C1 1000x400")
    ppm = canvas.to_ppm()
    assert ppm == O.to_ppm(canvas.data)
    assert ppm.startswith(b"P3\n1000 400\n255\n") and ppm.endswith(b"\n")
    assert max(len(line) for line in ppm.split(b"\n")) <= 70


def test_device_ppm_formatter(torch):
    """Canvas::to_ppm formatted on the device (next-1 of SURVEY.md 8(f)): byte-identical to the oracle's
    canvas.rs restatement for ragged sizes and edge values, and to the host writer on a full 4096^2 frame."""
    import time
    rng = np.random.default_rng(17)
    world, camera, depth = scenes.soft_shadows(64, 64)
    r = _renderer(world, camera)
    for (w, h) in [(1, 1), (5, 3), (10, 2), (23, 7), (70, 3), (101, 4), (64, 64), (333, 17)]:
        img = rng.uniform(-0.2, 1.3, (h, w, 3)).astype(f32)
        img[0, 0] = [np.nan, np.inf, -np.inf]
        if w > 4:
            img[-1, -4:] = 1.0   # rows ending in three-digit values exercise the wrap at the row end
        got = r.to_ppm(torch.from_numpy(img).cuda())
        assert got == O.to_ppm(img), (w, h)
    world, camera, depth = scenes.CONFIGS["C3"]()
    r = _renderer(world, camera)
    frame = r.render(depth)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    text = r.to_ppm(frame)
    t_dev = time.perf_counter() - t0
    host = P.Canvas(4096, 4096, frame.cpu().numpy())
    t0 = time.perf_counter()
    ref = host.to_ppm()
    t_host = time.perf_counter() - t0
    assert text == ref
    assert text.startswith(b"P3\n4096 4096\n255\n") and len(text) > 100_000_000
    print("\nto_ppm 4096^2: device formatter + D2H %.3f s, host writer %.3f s, %d bytes" % (t_dev, t_host, len(text)))


@pytest.mark.parametrize("name,size", [("first_scene", (4096, 2048)), ("first_plane", (4096, 2048)), ("first_patterns", (4096, 2048)),
                                       ("reflect_refract", (4096, 2048)), ("hexagons", (4096, 2048)), ("first_textures", (4096, 2048)),
                                       ("skybox", (4096, 2048)), ("grouped_grid", (4096, 4096)), ("mesh", (2048, 2048)),
                                       ("here_be_dragons", (4000, 1600)), ("soft_shadows", (1536, 1536))])
def test_demo_scenes_whole_frame_at_the_sizes_they_are_timed_at(torch, name, size):
    """DESIGN.md section 8 quotes a kernel time for each of these frames: the frame that was timed equals the reference's,
    pixel for pixel and ray for ray -- with whatever kernel family, lanes per pixel, block list, nodes, scene rectangle
    the library's default policy picks at that size (the small-frame tests pick those by switches)."""
    world, camera, depth = getattr(scenes, name)(*size)
    r = _renderer(world, camera)
    image = r.render(depth).cpu().numpy()
    st = r.stats()
    # ... and so do the later frames of the same scene, which is what the table times: block lists are made from the frames
    # before (rtc_device.hip refine_block_list)
    for _ in range(3):
        later = r.render(depth).cpu().numpy()
    later_rays = r.stats()["rays"]
    r.close()
    assert not np.isnan(image).any()
    assert np.array_equal(later, image) and later_rays == st["rays"], "%s: the fourth frame differs from the first" % name
    _check_whole_frame(image, st["rays"], world, camera, depth, "%s %dx%d" % ((name,) + size))
