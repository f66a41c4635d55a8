#!/usr/bin/env python3
"""Development: what the cell cones (rtc_kernel_core.h classify_cells) decide on a frame -- rays answered without a sample,
kernel time -- with the switch on and off, one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTC_AMD_LIB", os.path.join(ROOT, "ray_tracer_challenge_amd", "librtc_amd_dev.so"))


def main():
    from ray_tracer_challenge_amd import scenes
    from ray_tracer_challenge_amd.renderer import Renderer
    scene, w, h = (sys.argv[1:4] + ["soft_shadows", "4096", "4096"][len(sys.argv) - 1:])[:3]
    world, camera, depth = getattr(scenes, scene)(int(w), int(h))
    for label, env in (("cells on", {}), ("cells off", {"RTC_AMD_CELL_CULL": "0"}), ("cells on, fast off", {"RTC_AMD_FAST_SHADOW": "0"}),
                       ("cells off, fast off", {"RTC_AMD_CELL_CULL": "0", "RTC_AMD_FAST_SHADOW": "0"})):
        for k in ("RTC_AMD_CELL_CULL", "RTC_AMD_FAST_SHADOW"):
            os.environ.pop(k, None)
        os.environ.update(env)
        r = Renderer(world, camera, device=0)
        out = r.alloc()
        for _ in range(4):
            r.render(depth, out=out)
        r.stats()
        for _ in range(10):
            r.render(depth, out=out)
        st = r.stats()
        print("%-22s kernel %.4f ms  rays %d  without a sample %d (%.1f %%)  sampled %d  shade points %d" % (
            label, st["kernel_ms"], st["rays"], st["culled_shadow_rays"], 100.0 * st["culled_shadow_rays"] / st["rays"],
            st["rays"] - st["culled_shadow_rays"], st["shaded_hits"]))
        r.close()


if __name__ == "__main__":
    main()
