import os, sys, subprocess
sys.path.insert(0, "/root/repo")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import ray_tracer_challenge_amd as P
    from tests import helpers as H
    from tests.test_groups import _small_tree_world
    from ray_tracer_challenge_amd.renderer import Renderer
    seed = int(sys.argv[2])
    world, camera = _small_tree_world(seed, True)
    exp, rays = H.oracle_camera(camera).render(H.oracle_world(world), 4, threads=8)
    r = Renderer(world, camera, device=0)
    img = r.render(4).cpu().numpy()
    st = r.stats()
    bad = (img != exp).any(axis=2)
    ys, xs = np.nonzero(bad)
    print(seed, sys.argv[3], r.kernel_name, "bad px", int(bad.sum()), "rays", st["rays"], rays, "first", list(zip(xs[:6].tolist(), ys[:6].tolist())), flush=True)
else:
    base = dict(os.environ, RTC_AMD_SPECIALIZE="1", RTC_AMD_GATES="0")
    for seed in (0, 2):
        for name, env in (("default", {}), 
                          ("n12all", {"RTC_AMD_JIT_FLAGS": "-DRTC_DBG_N12_ALL"}), ("n12noprune", {"RTC_AMD_JIT_FLAGS": "-DRTC_DBG_N12_NOPRUNE"})):
            subprocess.run([sys.executable, __file__, "child", str(seed), name], env=dict(base, **env))
