"""Block lists made from the previous frame's wave times (rtc_device.hip refine_block_list; -m gpu).

The first frame of a scene with divided meshes under a point light is rendered from a block list that knows three kinds of
tile; that launch times its waves, and every later frame's list -- which tiles start first, how many lanes trace a pixel of
each -- is made from those times.  None of it may change a pixel or the ray count: every frame of a sequence is compared with
the oracle, whole, with the feedback on (first frame: the list as built; later ones: refined once, twice) and off, and for
a frame rendered as two interleaved partitions (each has a list and a refinement of its own).
"""
import os

import numpy as np
import pytest

from ray_tracer_challenge_amd import scenes
from tests import helpers as H

pytestmark = pytest.mark.gpu
THREADS = min(16, len(os.sched_getaffinity(0)))


def _renderer(world, camera):
    from ray_tracer_challenge_amd.renderer import Renderer
    return Renderer(world, camera, device=0)


@pytest.fixture
def feedback_env():
    saved = os.environ.get("RTC_AMD_BLOCK_FEEDBACK")

    def set_(v):
        if v is None:
            os.environ.pop("RTC_AMD_BLOCK_FEEDBACK", None)
        else:
            os.environ["RTC_AMD_BLOCK_FEEDBACK"] = v
    yield set_
    set_(saved)


@pytest.mark.parametrize("name,size", [("mesh", (320, 240)), ("mesh", (1024, 768)), ("here_be_dragons", (500, 200)),
                                       # regular grids (flat worlds, trees without mesh runs): the same 16 x 16 blocks, ordered
                                       ("glass_and_mirror", (520, 392)), ("soft_shadows", (600, 248)), ("reflect_refract", (648, 328)),
                                       ("hexagons", (600, 300)), ("first_textures", (512, 256)), ("sphere_grid", (1024, 1024)),
                                       # frames that share an area light's cells between lanes: a list of one lane count, re-cut by the feedback
                                       ("soft_shadows", (1000, 400)), ("patterns_medley", (256, 192)), ("groups_medley", (256, 192))])
def test_every_frame_of_a_sequence_equals_the_oracle(feedback_env, name, size):
    world, camera, depth = getattr(scenes, name)(*size)
    exp, exp_rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=THREADS)
    for mode in (None, "0"):  # the library's default (feedback on), and off
        feedback_env(mode)
        r = _renderer(world, camera)  # (switches are read when the context is created)
        for frame in range(5):
            image = r.render(depth).cpu().numpy()
            st = r.stats()
            what = "%s %dx%d feedback=%s frame %d" % (name, size[0], size[1], mode, frame)
            if not np.array_equal(image, exp):
                H.assert_images_equal(image, exp, what)
            assert st["rays"] == exp_rays, (what, st["rays"], exp_rays)
        r.close()


@pytest.mark.parametrize("name,size", [("mesh", (640, 480)), ("glass_and_mirror", (640, 480))])
def test_partitions_refine_their_own_lists(feedback_env, name, size):
    feedback_env(None)
    world, camera, depth = getattr(scenes, name)(*size)
    exp, exp_rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=THREADS)
    r = _renderer(world, camera)
    n_parts, band = 2, 48
    for frame in range(4):
        image = np.zeros_like(exp)
        rays = 0
        for part in range(n_parts):
            q = r.partition(band, n_parts, part)
            rows = r.render(depth, part=q).cpu().numpy()
            rays += r.stats()["rays"]
            cursor = 0
            for b in range(part, (camera.height + band - 1) // band, n_parts):
                y0, y1 = b * band, min(camera.height, (b + 1) * band)
                image[y0:y1] = rows[cursor:cursor + (y1 - y0)]
                cursor += y1 - y0
            assert cursor == rows.shape[0]
        if not np.array_equal(image, exp):
            H.assert_images_equal(image, exp, "two partitions, frame %d" % frame)
        assert rays == exp_rays, (frame, rays, exp_rays)
    r.close()


@pytest.mark.parametrize("name,size", [("mesh", (320, 240)), ("glass_and_mirror", (400, 300))])
def test_a_change_of_depth_between_frames(feedback_env, name, size):
    """What a block costs depends on the recursion depth: lists are kept per depth, and a frame at another depth is its depth's
    first frame."""
    feedback_env(None)
    world, camera, _ = getattr(scenes, name)(*size)
    expected = {d: H.oracle_camera(camera).render(H.oracle_world(world), d, threads=THREADS) for d in (5, 1, 0)}
    r = _renderer(world, camera)
    for frame, d in enumerate((5, 5, 5, 1, 1, 5, 0, 1, 5, 0)):
        image = r.render(d).cpu().numpy()
        exp, exp_rays = expected[d]
        if not np.array_equal(image, exp):
            H.assert_images_equal(image, exp, "%s frame %d at depth %d" % (name, frame, d))
        assert r.stats()["rays"] == exp_rays, (frame, d)
    r.close()


@pytest.mark.parametrize("size", [(2, 2), (17, 9), (16, 16), (33, 65), (130, 3), (257, 129)])
@pytest.mark.parametrize("name", ["soft_shadows", "glass_and_mirror", "mesh"])
def test_odd_and_tiny_frames(feedback_env, name, size):
    feedback_env(None)
    world, camera, depth = getattr(scenes, name)(*size)
    exp, exp_rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=THREADS)
    r = _renderer(world, camera)
    for frame in range(4):
        image = r.render(depth).cpu().numpy()
        if not np.array_equal(image, exp):
            H.assert_images_equal(image, exp, "%s %dx%d frame %d" % (name, size[0], size[1], frame))
        assert r.stats()["rays"] == exp_rays, (name, size, frame)
    r.close()


@pytest.mark.parametrize("name,size", [("mesh", (320, 240)), ("glass_and_mirror", (400, 304)), ("soft_shadows", (600, 248)), ("reflect_refract", (648, 328))])
def test_a_camera_that_moves(feedback_env, name, size):
    """An animation: rtc_ctx_set_scene with a new camera before every frame.  The block lists outlive the change of scene (same
    frame size: restart_block_lists) -- each frame runs from the list the frames before made, is timed, and re-cuts the next --
    and every frame is the new scene's image."""
    import ray_tracer_challenge_amd as P
    feedback_env(None)
    world, camera0, depth = getattr(scenes, name)(*size)
    r = _renderer(world, camera0)
    for frame in range(7):
        a = 0.03 * frame
        camera = P.Camera(size[0], size[1], scenes.PI / np.float32(3.0),
                          P.view_transform(P.point(0.2 + 3.0 * np.sin(a), 2.2, -5.5 * np.cos(a)), P.point(0, 0.8, 0), P.vector(0, 1, 0)))
        exp, exp_rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=THREADS)
        r.set_scene(world, camera)
        for again in range(2 if frame % 3 == 2 else 1):  # (now and then the same scene twice: the early return of set_scene)
            image = r.render(depth).cpu().numpy()
            if not np.array_equal(image, exp):
                H.assert_images_equal(image, exp, "%s frame %d.%d" % (name, frame, again))
            assert r.stats()["rays"] == exp_rays, (name, frame, again)
    r.close()


@pytest.mark.parametrize("name,size", [("soft_shadows", (600, 248)), ("mesh", (320, 240)), ("first_textures", (512, 256))])
def test_cameras_and_objects_that_move_in_turn(feedback_env, name, size):
    """Frames whose camera alone has moved keep the records that are resident (Renderer.set_camera; rtc_ctx_set_scene uploads
    nothing when the records are the same); frames in which an object has moved replace them -- in turn, every frame against
    the oracle."""
    import ray_tracer_challenge_amd as P
    feedback_env(None)
    world, camera, depth = getattr(scenes, name)(*size)
    r = _renderer(world, camera)
    mover = [o for o in world.objects if not isinstance(o, P.GroupShape)][-1]
    t0 = mover.transform.copy()
    try:
        for frame in range(8):
            if frame % 3 == 1:  # an object moves (the camera stays)
                mover.set_transformation(P.translation(0.05 * frame, 0.0, 0.02 * frame) @ t0)
                r.set_scene(world, camera)
            else:  # the camera moves (the world stays)
                a = 0.03 * frame
                camera = P.Camera(size[0], size[1], scenes.PI / np.float32(3.0),
                                  P.view_transform(P.point(0.2 + 3.0 * np.sin(a), 2.2, -5.5 * np.cos(a)), P.point(0, 0.8, 0), P.vector(0, 1, 0)))
                r.set_camera(camera)
            exp, exp_rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=THREADS)
            image = r.render(depth).cpu().numpy()
            if not np.array_equal(image, exp):
                H.assert_images_equal(image, exp, "%s frame %d" % (name, frame))
            assert r.stats()["rays"] == exp_rays, (name, frame)
    finally:
        mover.set_transformation(t0)
        r.close()


@pytest.mark.parametrize("name,size", [("mesh", (320, 240)), ("soft_shadows", (600, 248))])
def test_partitions_of_a_scene_that_changes(feedback_env, name, size):
    """Two interleaved partitions, each with a list of its own, under a camera that moves before every frame: every list is
    timed by its own launch's events and re-cut on its own account (restart_block_lists) -- every frame against the oracle."""
    import ray_tracer_challenge_amd as P
    feedback_env(None)
    world, camera, depth = getattr(scenes, name)(*size)
    r = _renderer(world, camera)
    n_parts, band = 2, 48
    for frame in range(9):
        a = 0.05 * frame
        camera = P.Camera(size[0], size[1], scenes.PI / np.float32(3.0),
                          P.view_transform(P.point(0.2 + 3.0 * np.sin(a), 2.2, -5.5 * np.cos(a)), P.point(0, 0.8, 0), P.vector(0, 1, 0)))
        exp, exp_rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=THREADS)
        r.set_camera(camera)
        image = np.zeros_like(exp)
        rays = 0
        for part in range(n_parts):
            q = r.partition(band, n_parts, part)
            rows = r.render(depth, part=q).cpu().numpy()
            rays += r.stats()["rays"]
            cursor = 0
            for b in range(part, (camera.height + band - 1) // band, n_parts):
                y0, y1 = b * band, min(camera.height, (b + 1) * band)
                image[y0:y1] = rows[cursor:cursor + (y1 - y0)]
                cursor += y1 - y0
        if not np.array_equal(image, exp):
            H.assert_images_equal(image, exp, "%s, two partitions, frame %d" % (name, frame))
        assert rays == exp_rays, (name, frame, rays, exp_rays)
    r.close()
