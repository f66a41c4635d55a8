import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ray_tracer_challenge_amd import scenes
world, camera, depth = scenes.soft_shadows(4096, 4096)
for pin in ("0", "1", "0", "1"):
    os.environ["RTC_AMD_PIN_OUTPUT"] = pin
    camera.render(world, depth)  # warm (JIT cache etc.)
    t = time.perf_counter(); c = camera.render(world, depth); dt = time.perf_counter() - t
    print("pin", pin, "rtc_render wall %.1f ms" % (dt * 1e3), "kernel %.2f ms" % camera.last_stats["kernel_ms"], "sum", float(c.data.sum()))
