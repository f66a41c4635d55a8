#!/usr/bin/env python3
"""Development: what makes a scene's FIRST frame slower than its later ones?  One fresh context per variant, the first three
frames' kernel times: as is; after a memory-bound warm-up (clocks); after a small render of the same scene in another context
(code object, instruction cache, TLB); with the frame-before feedback off (scheduling).

    python tools/first_frame_probe.py [scene width height]
"""
import os
import subprocess
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(scene, w, h, variant):
    import torch
    from ray_tracer_challenge_amd import scenes
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = getattr(scenes, scene)(w, h)
    if variant == "busy":
        x = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
        for _ in range(200):
            x.zero_()
        torch.cuda.synchronize()
    if variant == "valu":
        x = torch.randn(1 << 24, device="cuda")
        for _ in range(50):
            x = torch.sin(x) * 1.0001
        torch.cuda.synchronize()
    if variant == "other_ctx":
        r0 = Renderer(world, camera, device=0)
        r0.render(depth)
        r0.stats()
        r0.close()
    r = Renderer(world, camera, device=0)
    out = r.alloc()
    if variant == "touch_out":  # the frame's buffer written once before the first frame (its pages' translations, the memory-side cache)
        out.zero_()
        torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        r.render(depth, out=out)
        ts.append(r.stats()["kernel_ms"])
    print(json.dumps({"variant": variant, "kernel": r.kernel_name, "ms": ts}))


def main():
    scene, w, h = (sys.argv[1:4] + ["soft_shadows", "4096", "4096"][len(sys.argv) - 1:])[:3]
    for variant, env in (("cold", {}), ("busy", {}), ("valu", {}), ("other_ctx", {}), ("touch_out", {}), ("no_feedback", {"RTC_AMD_BLOCK_FEEDBACK": "0"})):
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, __file__, "child", scene, w, h, variant], env=e, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(line[-1] if line else (variant, p.stderr[-300:]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    else:
        main()
