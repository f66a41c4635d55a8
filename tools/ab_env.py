#!/usr/bin/env python3
"""A/B harness over environment switches (development tool, not part of the product).

    python tools/ab_env.py [--scene S --size W [--height H] --steps N --rounds R] 'name|ENV=value|ENV2=value with spaces' ...

Each variant renders the same frame in its own process (interleaved rounds on one box, every other round in reverse order) with its environment applied --
RTC_AMD_JIT_FLAGS='-DFOO=1', RTC_AMD_REG_LEVELS=3, ... -- and prints kernel ms, the kernel's name and an image hash,
so that a variant which changes a single output bit is caught at once.  A variant with no assignments is the default.
"""
import hashlib
import json
import os

# the development switches this tool drives exist only in the development build of the library
os.environ.setdefault("RTC_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ray_tracer_challenge_amd", "librtc_amd_dev.so"))
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(scene, w, h, steps):
    from ray_tracer_challenge_amd import scenes
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = getattr(scenes, scene)(w, h)
    r = Renderer(world, camera, device=0)
    out = r.alloc()
    for _ in range(5):
        r.render(depth, out=out)
    r.stats()
    for _ in range(steps):
        r.render(depth, out=out)
    st = r.stats()
    digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({"kernel_ms": st["kernel_ms"], "rays": st["rays"], "hash": digest, "kernel": r.kernel_name, "id": r.kernel_id,
                      "flags": st["flags"]}))


def child_first(scene, w, h, steps):
    """--first: every step is a scene's FIRST frame -- a fresh context each (the process's first one, unmeasured, pays the compile)."""
    from ray_tracer_challenge_amd import scenes
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = getattr(scenes, scene)(w, h)
    ms, st, digest, name, kid = 0.0, None, None, None, None
    for i in range(steps + 1):
        r = Renderer(world, camera, device=0)
        out = r.alloc()
        r.render(depth, out=out)
        st = r.stats()
        if i:
            ms += st["kernel_ms"]
        digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]
        name, kid = r.kernel_name, r.kernel_id
        r.close()
    print(json.dumps({"kernel_ms": ms / steps, "rays": st["rays"], "hash": digest, "kernel": name, "id": kid, "flags": st["flags"]}))


def main(args):
    scene, w, h, steps, rounds = "soft_shadows", 4096, 0, 10, 3
    mode = "child"
    variants = []
    it = iter(args)
    for a in it:
        if a == "--scene": scene = next(it)
        elif a == "--size": w = int(next(it))
        elif a == "--height": h = int(next(it))
        elif a == "--steps": steps = int(next(it))
        elif a == "--rounds": rounds = int(next(it))
        elif a == "--first": mode = "child_first"
        else:
            parts = a.split("|")
            variants.append((parts[0], dict(p.split("=", 1) for p in parts[1:])))
    h = h or w
    res = {n: [] for n, _ in variants}
    info = {}
    for rnd in range(rounds):
        # (every other round in reverse: the later process of a round runs on a warmer chip, 1 - 1.5 % faster -- profiles/r03_ab_fewer_compares.txt)
        for name, env_extra in (variants if rnd % 2 == 0 else variants[::-1]):
            env = dict(os.environ)
            env.update(env_extra)
            p = subprocess.run([sys.executable, __file__, mode, scene, str(w), str(h), str(steps)], env=env, capture_output=True, text=True)
            if p.returncode != 0:
                print(name, "FAILED", p.stderr[-600:])
                continue
            d = json.loads(p.stdout.strip().splitlines()[-1])
            res[name].append(d["kernel_ms"])
            info[name] = d
    ref = info.get(variants[0][0])
    print("%s %dx%d, %d steps x %d rounds" % (scene, w, h, steps, rounds))
    for name, _ in variants:
        if res[name]:
            v = sorted(res[name])
            d = info[name]
            same = ref and (d["hash"], d["rays"]) == (ref["hash"], ref["rays"])
            print("%-28s kernel_ms min %.4f med %.4f  hash %s rays %d %s%s  %s" % (
                name, v[0], v[len(v) // 2], d["hash"], d["rays"], "" if same else "<-- DIFFERS ", "JIT-FALLBACK " if d["flags"] else "", d["kernel"]))


if __name__ == "__main__":
    if sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    elif sys.argv[1] == "child_first":
        child_first(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    else:
        main(sys.argv[1:])
