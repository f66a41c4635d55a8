"""GPU tests at BASELINE.json's full sizes (-m gpu).

The single-threaded oracle cannot render 16.7 M pixels x ~100 rays in a test,
so full-size parity is established by (a) whole rows sampled from the frame and
compared bit-for-bit with the oracle, and (b) size-independent properties:
partition invariance (any band split assembles to the identical image and the
identical ray count), the never-traced last row/column, determinism, and the
device-side quantiser against the oracle's scale_color.
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


def _check_rows(image, world, camera, depth, rows):
    ow, oc = H.oracle_world(world), H.oracle_camera(camera)
    for y in rows:
        exp, _ = oc.render(ow, depth, threads=THREADS, rows=(y, y + 1))
        H.assert_images_equal(image[y:y + 1], exp[y:y + 1], "row %d" % y)


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
    _check_rows(img, world, camera, depth, [0, 1400, 2300, 2700, 3333, 4094])
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


def test_c3_constant_jitter_rows(torch):
    world, camera, depth = scenes.soft_shadows(4096, 4096, jitter=("constant", 0.5))
    r = _renderer(world, camera)
    img = r.render(depth).cpu().numpy()
    _check_rows(img, world, camera, depth, [2048, 2650, 3500])


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
    _check_rows(img, world, camera, depth, [100, 1800, 2048, 2400, 3000, 4000])
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
    _check_rows(whole, world, camera, depth, [1000, 4096, 6000])
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
    """BASELINE config 1 (the demo binary's 1000x400): sampled rows + PPM header / size sanity."""
    world, camera, depth = scenes.CONFIGS["C1"]()
    canvas = camera.render(world, depth)
    _check_rows(canvas.data, world, camera, depth, [50, 200, 300, 398])
    ppm = canvas.to_ppm()
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
