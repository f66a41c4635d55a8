"""Python counterpart of the reference's demo binaries (demos/src/bin/*.rs): builds the demo's scene, renders it on the
GPU and prints the P3 PPM to stdout exactly as the binaries do (`println!("{}", canvas.to_ppm())`: the PPM text plus
one more newline), timing to stderr like camera.rs:79-89.

    python -m ray_tracer_challenge_amd.demo soft_shadows > soft_shadows.ppm
    python -m ray_tracer_challenge_amd.demo here_be_dragons dragon.obj --size 1000x400 > dragons.ppm
    python -m ray_tracer_challenge_amd.demo first_textures earth.ppm > textures.ppm

The three demos that read a file (`argv[1]`: here_be_dragons.rs:38, first_textures.rs, skybox.rs) take it as the
second argument; without it a procedural stand-in is used (the files are not in the reference's repository)."""
import argparse
import sys
import time

from . import scenes

DEMOS = {  # name -> (scene function, default size as in the demo's CANVAS_WIDTH / CANVAS_HEIGHT, file keyword)
    "soft_shadows": (scenes.soft_shadows, (1000, 400), None),
    "first_scene": (scenes.first_scene, (1000, 500), None),
    "first_plane": (scenes.first_plane, (100, 50), None),
    "first_patterns": (scenes.first_patterns, (1000, 500), None),
    "reflect_refract": (scenes.reflect_refract, (1000, 500), None),
    "hexagons": (scenes.hexagons, (1000, 500), None),
    "first_textures": (scenes.first_textures, (1000, 500), "earth_ppm"),
    "skybox": (scenes.skybox, (800, 400), None),
    "here_be_dragons": (scenes.here_be_dragons, (1000, 400), "obj_text"),
}


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m ray_tracer_challenge_amd.demo", description=__doc__.split("\n\n")[0])
    ap.add_argument("demo", choices=sorted(DEMOS))
    ap.add_argument("file", nargs="?", help="the demo's argv[1] (OBJ mesh / PPM texture)")
    ap.add_argument("--size", help="WxH (default: the demo's own)")
    ap.add_argument("--out", help="write the PPM here instead of stdout")
    a = ap.parse_args(argv)
    fn, size, file_kw = DEMOS[a.demo]
    if a.size:
        size = tuple(int(v) for v in a.size.lower().split("x"))
    kw = {}
    if a.file:
        if not file_kw:
            ap.error("%s takes no file" % a.demo)
        with open(a.file, "r") as f:
            kw[file_kw] = f.read()
    t0 = time.time()
    world, camera, depth = fn(size[0], size[1], **kw)
    t1 = time.time()
    canvas = camera.render(world, depth)
    ppm = canvas.to_ppm()
    t2 = time.time()
    st = camera.last_stats
    print("scene built in %.3f s; rendered %dx%d (%d rays, kernel %.3f ms) and formatted in %.3f s"
          % (t1 - t0, size[0], size[1], st["rays"], st["kernel_ms"], t2 - t1), file=sys.stderr)
    if a.out:
        with open(a.out, "wb") as f:
            f.write(ppm + b"\n")
    else:
        sys.stdout.buffer.write(ppm + b"\n")
        sys.stdout.buffer.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main())
