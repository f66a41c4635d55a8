#!/usr/bin/env python3
"""development: what ONE rtc_render_ex call costs a fresh process (the reference renders one frame per process, camera.rs:76) --
with an empty JIT cache, with a filled one, and with the ahead-of-time kernels (RTC_AMD_SPECIALIZE=0).

    python tools/first_call.py [scene width height]
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(scene, w, h):
    import time
    import numpy as np
    import ray_tracer_challenge_amd as P
    from ray_tracer_challenge_amd import scenes
    world, camera, depth = getattr(scenes, scene)(w, h)
    buf = np.zeros((camera.height, camera.width, 3), dtype=np.uint8)
    P.lib().rtc_device_count()  # (the HIP runtime's own start-up is not the library's: taken out of the call's time)
    extra = {}
    if os.environ.get("FIRST_CALL_CONTEXT_FIRST"):  # ... and the device's context and first queue, created here by a persistent context that is dropped again
        import ctypes as C
        t0 = time.perf_counter()
        ctx = C.c_void_p()
        P.lib().rtc_ctx_create(0, C.byref(ctx))
        P.lib().rtc_ctx_destroy(ctx)
        extra["ctx_create_destroy_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        camera.render(world, depth, quantize=True, out=buf)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"calls_ms": [round(t, 2) for t in ts], "kernel_ms": round(camera.last_stats["kernel_ms"], 4), "flags": camera.last_stats["flags"], **extra}))


def main():
    scene, w, h = (sys.argv[1:4] + ["soft_shadows", "4096", "4096"][len(sys.argv) - 1:])[:3]
    cache = tempfile.mkdtemp(prefix="rtc_first_call_")
    try:
        for name, env in (("empty cache", {"RTC_AMD_JIT_CACHE": cache}), ("filled cache", {"RTC_AMD_JIT_CACHE": cache}),
                          ("ahead-of-time kernels", {"RTC_AMD_SPECIALIZE": "0"}),
                          ("filled cache, device context made first", {"RTC_AMD_JIT_CACHE": cache, "FIRST_CALL_CONTEXT_FIRST": "1"})):
            e = dict(os.environ)
            e.update(env)
            p = subprocess.run([sys.executable, __file__, "child", scene, w, h], env=e, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            print("%-42s %s" % (name, line[-1] if line else p.stderr[-300:]))
    finally:
        shutil.rmtree(cache, ignore_errors=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        main()
