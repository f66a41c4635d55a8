"""Both compilers a process may hold (-m gpu).  A Python process that imported torch first resolves the wheel's bundled
libhiprtc / libamd_comgr; a caller that links librtc_amd.so without torch -- the reference's Rust host, the C++ demos -- and
any process started under rocprofv3 get the system's (/opt/rocm/lib).  One hiprtcVersion, different code objects for the
same kernel source (LABNOTES "Round 4": the kernel id carries the code object's checksum for that reason).  The rest of the
suite runs with the wheel's; this renders a scene of each kernel family in a child process that holds the system's, with
the disk cache off so that it compiles for itself, and compares every pixel and the ray count with the oracle."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %r)
from ray_tracer_challenge_amd import scenes
from ray_tracer_challenge_amd.renderer import Renderer
from tests import helpers as H
out = []
for name, size in (("soft_shadows", (600, 248)), ("glass_and_mirror", (520, 392)), ("reflect_refract", (648, 328)), ("first_textures", (512, 256)),
                   ("hexagons", (600, 300)), ("mesh", (640, 480)), ("sphere_grid", (1024, 1024))):
    world, camera, depth = getattr(scenes, name)(*size)
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), depth, threads=min(16, len(os.sched_getaffinity(0))))
    r = Renderer(world, camera, device=0)
    for frame in range(2):
        img = r.render(depth).cpu().numpy()
        st = r.stats()
        same = bool(((img == exp) | (np.isnan(img) & np.isnan(exp))).all())
        out.append({"scene": name, "frame": frame, "kernel": r.kernel_name, "id": r.kernel_id, "same": same, "rays": st["rays"] == rays})
    r.close()
print("RESULT " + json.dumps(out))
""" % ROOT


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    env["RTC_AMD_JIT_CACHE"] = "0"     # every process compiles for itself
    env["RTC_AMD_SPECIALIZE"] = "1"    # scene-compiled kernels whatever the frame's size
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, p.stdout[-2000:]
    return json.loads(line[-1][7:])


def test_the_system_compilers_code_objects_render_the_same_frames():
    if not os.path.exists("/opt/rocm/lib/libhiprtc.so"):
        pytest.skip("no system hiprtc beside the wheel's")
    system = _run({"LD_LIBRARY_PATH": "/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", "")})
    for rec in system:
        assert rec["same"] and rec["rays"], rec
    wheel = _run({})
    for rec in wheel:
        assert rec["same"] and rec["rays"], rec
    # (informational: where the two compilers emit different code the ids differ in their second half, never in their first)
    for a, b in zip(system, wheel):
        assert a["id"].split(".")[0] == b["id"].split(".")[0], (a, b)
