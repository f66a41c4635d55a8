#!/usr/bin/env python3
"""A/B harness for kernel variants (development tool, not part of the product).

Local:   python tools/ab.py build  tagA="-DFOO=1" tagB="-fno-slp-vectorize" ...
GPU box: python tools/ab.py run [--scene soft_shadows --size 4096 --steps 8] tagA tagB ...
Each variant is built to ray_tracer_challenge_amd/variants/librtc_amd_<tag>.so; `run` renders the same frame
with each (interleaved rounds, one process per variant per round) and prints kernel ms + an image hash so that
any variant that changes a single output bit is caught immediately.
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "ray_tracer_challenge_amd", "variants")


def build(specs):
    from ray_tracer_challenge_amd import build as B
    os.makedirs(VDIR, exist_ok=True)
    procs = []
    for spec in specs:
        tag, _, flags = spec.partition("=")
        out = os.path.join(VDIR, "librtc_amd_%s.so" % tag)
        cmd = ["hipcc"] + B.FLAGS + flags.split() + ["-o", out] + B.SOURCES + ["-lhiprtc", "-ldl"]
        procs.append((tag, subprocess.Popen(cmd)))
    for tag, p in procs:
        if p.wait() != 0:
            raise SystemExit("build of %s failed" % tag)
        print("built", tag)


def child(scene, size, steps):
    import torch
    from ray_tracer_challenge_amd import scenes
    from ray_tracer_challenge_amd.renderer import Renderer
    world, camera, depth = getattr(scenes, scene)(size, size)
    r = Renderer(world, camera, device=0)
    out = r.alloc()
    for _ in range(2):
        r.render(depth, out=out)
    r.stats()
    for _ in range(steps):
        r.render(depth, out=out)
    st = r.stats()
    h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({"kernel_ms": st["kernel_ms"], "rays": st["rays"], "hash": h}))


def run(args):
    scene, size, steps, rounds = "soft_shadows", 4096, 8, 3
    tags = []
    it = iter(args)
    for a in it:
        if a == "--scene": scene = next(it)
        elif a == "--size": size = int(next(it))
        elif a == "--steps": steps = int(next(it))
        elif a == "--rounds": rounds = int(next(it))
        else: tags.append(a)
    res = {t: [] for t in tags}
    hashes = {}
    for _ in range(rounds):
        for t in tags:
            env = dict(os.environ)
            env["RTC_AMD_LIB"] = os.path.join(VDIR, "librtc_amd_%s.so" % t) if t != "default" else ""
            p = subprocess.run([sys.executable, __file__, "child", scene, str(size), str(steps)], env=env,
                               capture_output=True, text=True)
            if p.returncode != 0:
                print(t, "FAILED", p.stderr[-400:])
                continue
            d = json.loads(p.stdout.strip().splitlines()[-1])
            res[t].append(d["kernel_ms"])
            hashes[t] = (d["hash"], d["rays"])
    ref = hashes.get(tags[0])
    for t in tags:
        if res[t]:
            v = sorted(res[t])
            print("%-24s kernel_ms min %.4f med %.4f  hash %s rays %d %s" % (
                t, v[0], v[len(v) // 2], hashes[t][0], hashes[t][1], "" if hashes[t] == ref else "  <-- DIFFERS"))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        run(sys.argv[2:])
