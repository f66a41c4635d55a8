#!/usr/bin/env python3
"""Where a one-call render's time goes (development tool): rtc_render_ex at 4096^2 into page-locked memory, u8 and f32, with
the development library's RTC_AMD_SEAM_TRACE=1 breakdown on stderr.   python tools/seam_trace.py [scene] [size]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RTC_AMD_LIB", os.path.join(ROOT, "ray_tracer_challenge_amd", "librtc_amd_dev.so"))
os.environ["RTC_AMD_SEAM_TRACE"] = "1"
sys.path.insert(0, ROOT)
import ray_tracer_challenge_amd as P  # noqa: E402
from ray_tracer_challenge_amd import _lib as L, scenes  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "soft_shadows"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
world, camera, depth = getattr(scenes, scene)(size, size)
lib, cs = P.lib(), world._c()
p = lib.rtc_host_alloc(size * size * 12)
dev = (C.c_int32 * 1)(0)
st = L.rtc_stats()
for name, q in (("f32 pinned", 0), ("u8 pinned", 1)):
    opts = L.rtc_opts(dev, 1, 0, q, 0)
    for i in range(6):
        t0 = time.perf_counter()
        L.check(lib.rtc_render_ex(C.byref(cs.scene), C.byref(camera._cam), depth, C.byref(opts), C.c_void_p(p), C.byref(st)))
        print("%s call %d: %.3f ms wall (python), kernel %.3f ms" % (name, i, (time.perf_counter() - t0) * 1e3, st.kernel_ms), file=sys.stderr)
lib.rtc_host_free(p)
