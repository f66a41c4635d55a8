#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on MI355X.

A "step" is one Camera::render of BASELINE.json's metric configuration: the
soft_shadows demo scene (area light 10x10, 4 objects, depth 5) at 4096x4096
("C3"), with the pinned hashed jitter.  On N GPUs (one process per GPU) the
image's 64-row bands are dealt round-robin to the ranks, each rank renders its
bands with the HIP kernel, and the bands are gathered to rank 0 over RCCL --
strong scaling: the job is one image whatever N is.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  value = rays traced by all ranks / wall time (the reference's count: every
World::intersect evaluation, including the area-light shadow rays the light-cone cull answers without an object test;
`tested_rays_per_s` is the rate of the rays that were tested).  `roofline` names the roof that binds this kernel --
FP32 VALU issue: VALU wave-instructions x 64 lanes per launch (PMC-measured, taken from the profiles/*_pmc.json whose
kernel id and workload match the kernel that just ran, else null) / the launch's mean HIP-event duration, against
256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 Tlane-op/s; `traffic` = PMC HBM bytes per launch from the same summary.
`contract_hbm_figure` is the contractual number of SURVEY.md 8(d) (algorithmic bytes = rays x n_objects x 64 B + 48 B
per shaded hit + 12 B per pixel over 8 TB/s): the scene is register-resident, so it is NOT physical traffic and
exceeds 1.  `one_shot` times the drop-in seam itself (rtc_render_ex: one call, host buffer out).  `cpu_baseline` is
the CPU oracle timed on a bounded row sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12  # 78.6: one non-FMA f32 lane-op per SIMD lane per cycle at the 2.4 GHz peak clock


# BASELINE.json's configurations by name (C1 .. C5) -> (scene function, width, height)
WORKLOADS = {
    "C1": ("soft_shadows", 1000, 400),       # the demo at its default resolution
    "C2": ("single_sphere", 1024, 1024),     # one sphere, point light: primary + shadow rays only
    "C3": ("soft_shadows", 4096, 4096),      # the metric configuration
    "C4": ("glass_and_mirror", 4096, 4096),  # glass sphere on a mirror plane, depth 5
    "C5": ("sphere_grid", 8192, 8192),       # 64 spheres, 8192^2: the configuration BASELINE tiles over 8 GPUs
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="a BASELINE.json configuration by name "
                    "(default C3, the metric configuration); --scene / --size / --height select anything else")
    ap.add_argument("--size", type=int, default=0, help="image edge in pixels (metric config: 4096)")
    ap.add_argument("--height", type=int, default=0, help="image height if not square (profiling other scenes: 1000x400 demos)")
    ap.add_argument("--scene", default=None, help="scene function in ray_tracer_challenge_amd.scenes")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU oracle sample (0 = skip)")
    ap.add_argument("--no-verify", action="store_true", help="skip the sampled-row parity check before timing")
    ap.add_argument("--no-one-shot", action="store_true", help="skip timing the one-call seam (rtc_render_ex)")
    ap.add_argument("--extra-parts", type=int, default=-1, help="N>1: parts rendered by rank 0 on top of its own in an "
                    "(N+E)-way band split; -1 = from the measured gather/render ratio, 0 = even split")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not collect the roofline's counters live (three short child runs of this "
                    "script under `rocprofv3 --pmc`, one counter group each); quote them from the committed profiles/*_pmc.json whose "
                    "kernel id matches instead")
    ap.add_argument("--force-dist", action="store_true", help="N=1: run the multi-GPU code path all the same -- init_process_group, the "
                    "band gather through dist.gather on the communication stream, its calibration and the u8 wire pass -- with ONE "
                    "rank (proves on a one-GPU box that RCCL loads, a communicator is created and the stream hand-off is right)")
    ap.add_argument("--wire", default="f32", choices=["f32", "u8"], help="N>1: what travels to rank 0 in the measured steps: the f32 "
                    "Canvas rows (12 B/pixel, the default) or the bytes Canvas::to_ppm prints (3 B/pixel, scale_color on the device); "
                    "the other format is timed beside it")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="N>1 started without a launcher: seconds the "
                    "parent waits for its ranks before ending them")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    # the workload: a named BASELINE configuration, overridden field by field by --scene / --size / --height
    scene, width, height = WORKLOADS[args.workload or "C3"]
    if args.scene is not None and args.workload is None and args.scene != scene:
        width = height = 4096  # (a scene chosen by name alone keeps the metric size, as before)
    if args.scene is not None:
        scene = args.scene
    if args.size:
        width = height = args.size
    if args.height:
        height = args.height
    args.scene, args.width, args.height_px = scene, width, height
    args.workload_name = next((k for k, v in WORKLOADS.items() if v == (scene, width, height)), None)
    return args


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it (no WORLD_SIZE / RANK in the environment): start the N
    ranks ourselves -- one child process per GPU, rendezvous on 127.0.0.1 -- and wait for them.  The parent makes NO GPU
    call (it imports neither torch nor the render library: a process that has initialised the GPU must not hand its
    work to children it then execs or outlives), relays nothing but what the children print themselves (rank 0 prints the
    JSON line on the stdout they inherit), and exits non-zero if any rank does -- ending the others first.
    The launcher-provided form (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) is untouched."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n = args.gpus
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RTC_BENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    deadline = time.monotonic() + args.launch_timeout
    code = 0
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 1
                    print("bench.py: rank %d exited with %d; ending the other ranks" % (r, rc), file=sys.stderr, flush=True)
            if code != 0 or time.monotonic() > deadline:
                if code == 0:
                    code = 124
                    print("bench.py: ranks still running after %.0f s; ending them" % args.launch_timeout, file=sys.stderr, flush=True)
                break
            if live:
                time.sleep(0.05)
    finally:
        for p in procs:  # exactly the processes started here, by pid
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return code


def cpu_baseline(world, camera, depth, budget_s):
    """Times the CPU oracle (a port of the reference's serial loop) on sampled row blocks of the same frame."""
    from oracle import oracle as O
    from tests import helpers as H
    # a 1-GPU box gives this job a CPU share of 16 threads whatever the affinity mask says
    cores = min(16, len(os.sched_getaffinity(0)))
    ow, oc = H.oracle_world(world), H.oracle_camera(camera)
    h = camera.height
    # calibrate on one row in the busy part of the image, single thread
    t0 = time.perf_counter()
    _, rays1 = oc.render(ow, depth, threads=1, rows=(h * 5 // 8, h * 5 // 8 + 1))
    t1 = time.perf_counter() - t0
    single = rays1 / t1 / 1e6 if t1 > 0 else 0.0
    # row blocks spread evenly over the image so the sample sees sky, floor and spheres in proportion
    rows_affordable = max(cores, int(budget_s / max(t1, 1e-6) * cores * 0.8))
    n_blocks = 16
    block = max(1, min(h // n_blocks, rows_affordable // n_blocks))
    total_rays, total_t, sampled = 0, 0.0, 0
    for b in range(n_blocks):
        y0 = (h * b) // n_blocks + (h // n_blocks - block) // 2
        t0 = time.perf_counter()
        _, r = oc.render(ow, depth, threads=cores, rows=(y0, y0 + block))
        total_t += time.perf_counter() - t0
        total_rays += r
        sampled += block
    return {
        "value": round(total_rays / total_t / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": "%d rows (16 blocks of %d rows spread over the %dx%d frame), %d rays, %.1f s wall; "
                  "row-parallel C++ oracle (no per-ray heap allocation or dyn dispatch: an upper bound on the Rust "
                  "reference)" % (sampled, block, camera.width, h, total_rays, total_t),
        "single_thread_value": round(single, 3),
    }


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # no launcher around us: be the launcher (before anything that could touch a GPU is even imported)
        raise SystemExit(launch_ranks(args, argv))
    import torch
    import torch.distributed as dist

    import ray_tracer_challenge_amd as P
    from ray_tracer_challenge_amd import scenes
    from ray_tracer_challenge_amd.dist import BandGather
    from ray_tracer_challenge_amd.renderer import Renderer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start `python bench.py --gpus %d` without a launcher (it starts its own "
                         "ranks) or with torch.distributed.run --nproc-per-node %d" % (args.gpus, world_size, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= n_dev:
        raise SystemExit("rank %d has no GPU (%d visible): RCCL needs one device per rank" % (local_rank, n_dev))
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist_on = world_size > 1 or args.force_dist  # the collectives run (with --force-dist also among one rank)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # --force-dist without a launcher: a rendezvous of one
            import socket
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            s_.close()
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    world, camera, depth = getattr(scenes, args.scene)(args.width, args.height_px)
    renderers = {}  # one context per part this rank renders: each keeps the counters and event timings of its own launches

    def renderer_for(p):
        if p not in renderers:
            renderers[p] = Renderer(world, camera, device=dev_index)
        return renderers[p]

    def make_runner(g, quantise=False):
        """K frames, software-pipelined: frame i's band gather (comm stream) overlaps frame i+1's render.  With
        `quantise` the rows travel as the bytes Canvas::to_ppm prints (device scale_color, canvas.rs:39-43)."""
        parts = g.parts()
        pdesc = {p: Renderer.partition(64, g.n_parts, p) for p in parts}
        scratch = None
        if quantise:
            scratch = {(s_, p): torch.empty(tuple(g.local_view(s_, p).shape), dtype=torch.float32, device=device)
                       for s_ in range(2) for p in parts}

        def run(n_steps):
            image = None
            for k in range(n_steps):
                slot = k % 2
                for p in parts:
                    r = renderer_for(p)
                    if quantise:
                        r.render(depth, out=scratch[(slot, p)], part=pdesc[p])
                        r.quantize(scratch[(slot, p)], out=g.local_view(slot, p))
                    else:
                        r.render(depth, out=g.local_view(slot, p), part=pdesc[p])
                g.start(slot)
                if k > 0:
                    image = g.finish((k - 1) % 2)
            if n_steps > 0:
                image = g.finish((n_steps - 1) % 2)
            return image
        return run

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def drain():
        """-> this rank's counters summed over its parts' last launches, and its kernel time per frame (the parts'
        mean launch times added up: they run back to back on one stream)."""
        tot = {"rays": 0, "shaded_hits": 0, "pixels": 0, "culled_shadow_rays": 0, "kernel_ms": 0.0}
        for r in renderers.values():
            st_ = r.stats()
            if st_["launches"]:
                for key in tot:
                    tot[key] += st_[key]
        return tot

    wire_u8 = args.wire == "u8" and dist_on  # the measured steps move bytes instead of f32 rows
    wire_dtype = torch.uint8 if wire_u8 else torch.float32
    gather = BandGather(camera.height, camera.width, 3, wire_dtype, device, rank, world_size, force_collective=args.force_dist)
    assert gather.local_view(0).shape[0] == renderer_for(rank).rows(Renderer.partition(64, world_size, rank))
    run = make_runner(gather, quantise=wire_u8)
    # the first frame of a scene is rendered in image order; from the next one on the blocks start in the order of the work
    # the frame before counted in them (rtc_device.hip refine_block_list).  Every ray is traced in every frame all the same.
    first = None
    if args.warmup > 0:
        run(1)
        fence()
        first = drain()
    run(max(0, args.warmup - 1))
    fence()
    warm = drain()
    if args.warmup <= 1 and first is not None:
        warm = first

    # xGMI is point-to-point: every peer's rows reach rank 0 over that peer's one link.  When moving a peer's share
    # takes longer than rendering it, rank 0 -- whose rows never travel -- takes E extra parts of an (N + E)-way split
    # (dist.BandGather).  E comes from the measured ratio of an un-overlapped gather to a render of one even share.
    extra_parts, calib = 0, None
    if dist_on and args.steps > 0:
        t0 = time.perf_counter()
        for _ in range(3):
            gather.start(0)
            gather.finish(0)
            torch.cuda.synchronize()
        t_gather = (time.perf_counter() - t0) / 3 * 1e3
        both = torch.tensor([warm["kernel_ms"], t_gather], dtype=torch.float64, device=device)
        dist.all_reduce(both, op=dist.ReduceOp.MAX)  # every rank derives the same E from the same two numbers
        k_ms, g_ms = float(both[0].item()), float(both[1].item())
        if args.extra_parts >= 0:
            extra_parts = args.extra_parts
        elif k_ms > 0:
            extra_parts = max(0, min(6, int(round(g_ms / k_ms)) - 1))
        if world_size == 1:
            extra_parts = 0  # (--force-dist: one rank has nobody to take parts from)
        calib = {"render_ms_even_share": round(k_ms, 4), "gather_ms_even_share_unoverlapped": round(g_ms, 4)}
        if extra_parts > 0:
            # the even split, timed the same way as the split `value` is measured with: so that the first run on real xGMI
            # links can be read (is the (N + E)-way split what the calibration promised?)
            fence()
            drain()
            t_even = time.perf_counter()
            run(args.steps)
            fence()
            e_even = torch.tensor([time.perf_counter() - t_even], dtype=torch.float64, device=device)
            dist.all_reduce(e_even, op=dist.ReduceOp.MAX)
            calib["even_split_ms_per_step"] = round(float(e_even.item()) / args.steps * 1e3, 4)
            drain()
            gather = BandGather(camera.height, camera.width, 3, wire_dtype, device, rank, world_size, extra_parts=extra_parts, force_collective=args.force_dist)
            run = make_runner(gather, quantise=wire_u8)
            run(max(2, min(args.warmup, 3)))
        fence()
        drain()

    t0 = time.perf_counter()
    image = run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0

    st = drain()  # counters of the last frame's launches + kernel time per frame on this rank
    st["kernel_ms"] = st["kernel_ms"] if st["kernel_ms"] > 0 else 0.0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    counts = torch.tensor([st["rays"], st["shaded_hits"], st["pixels"], st["culled_shadow_rays"]], dtype=torch.float64,
                          device=device)
    kern = torch.tensor([st["kernel_ms"]], dtype=torch.float64, device=device)
    if dist_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(kern, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    rays, shaded, pixels, culled = (int(v) for v in counts.tolist())
    renderer = renderer_for(rank)

    # N > 1, informational: the same K frames delivered to rank 0 as the bytes Canvas::to_ppm would print -- a quarter
    # of the xGMI traffic of the f32 Canvas rows that `value` is measured with.  Never fatal: a failure is reported
    # in the field instead.
    wire = None
    if dist_on and args.steps > 0:
        try:
            other = torch.float32 if wire_u8 else torch.uint8  # the format the measured steps did NOT use
            gather8 = BandGather(camera.height, camera.width, 3, other, device, rank, world_size, force_collective=args.force_dist)
            run8 = make_runner(gather8, quantise=not wire_u8)
            run8(2)
            fence()
            drain()
            t1 = time.perf_counter()
            img8 = run8(args.steps)
            fence()
            e8 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            dist.all_reduce(e8, op=dist.ReduceOp.MAX)
            drain()
            ok8 = None
            if rank == 0 and image is not None and img8 is not None:
                f32_img, u8_img = (img8, image) if wire_u8 else (image, img8)
                ok8 = bool(torch.equal(u8_img, renderer.quantize(f32_img.contiguous())))  # the bytes ARE the quantised f32 frame
                if wire_u8:
                    image = img8  # the frame whose rows are compared with the oracle below
            wire = {"encoding": ("f32 Canvas rows" if wire_u8 else "u8 (scale_color on the device)") + ", even split",
                    "ms_per_step": round(float(e8.item()) / args.steps * 1e3, 4),
                    "value": round(rays / (float(e8.item()) / args.steps) / 1e6, 2), "unit": "Mrays/s",
                    "gathered_bytes_per_step": int(camera.height * camera.width * (12 if wire_u8 else 3) * (world_size - 1) // world_size),
                    "equals_quantised_f32_frame": ok8}
        except Exception as exc:  # noqa: BLE001
            wire = {"error": repr(exc)[:200]}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        n_obj = len(world._c().leaves)  # leaf shapes (GroupShapes flattened)
        # roofline of the dominant kernel (render_kernel) on THIS rank: per-launch algorithmic bytes / mean duration
        algo_bytes = st["rays"] * n_obj * 64 + st["shaded_hits"] * 48 + st["pixels"] * 12
        achieved = algo_bytes / (st["kernel_ms"] * 1e-3) / 1e9 if st["kernel_ms"] > 0 else 0.0
        pmc = None
        if world_size == 1 and not args.no_live_pmc and not args.force_dist:
            lp = live_pmc(args, renderer.kernel_id)
            if lp is not None:
                pmc = ("live", lp)
        if pmc is None and world_size == 1:
            pmc = matching_pmc_summary(renderer.kernel_id, args.scene, camera.width, camera.height)
        verify = None
        if not args.no_verify and image is not None and image.dtype == torch.float32:
            verify = verify_rows(image, world, camera, depth)  # (the GATHERED frame when the collectives ran)
        line = {
            "metric": "Mrays/s", "value": round(rays / (elapsed / args.steps) / 1e6, 2), "unit": "Mrays/s",
            "mpixels_per_s": round(pixels / (elapsed / args.steps) / 1e6, 3),
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text(args, camera),
                       "workload_key": workload_key(args.scene, camera.width, camera.height),
                       "rays_per_frame": rays, "shaded_hits_per_frame": shaded, "pixels_per_frame": pixels,
                       # of rays_per_frame: area-light shadow rays whose answer followed from the conservative
                       # light-cone cull (no object test needed); they are counted because the reference casts them
                       "shadow_rays_resolved_by_light_cone_cull": culled,
                       "partition": "64-row bands round-robin over %d part(s)%s" % (
                           world_size + extra_parts, (", rank 0 renders %d of them; RCCL gather of f32 rows to rank 0"
                                                      % (extra_parts + 1)) if world_size > 1 else "")},
            # three significant digits: how many rays the cull answers is decided wave by wave (votes over 64 lanes), and which pixels
            # share a wave follows the frame's schedule -- the count moves in its fourth digit from run to run, the image and `rays` never
            "tested_rays_per_s": float("%.3g" % ((rays - culled) / (elapsed / args.steps))),
            "schedule": {"blocks": "16x16-pixel blocks; first frame of a scene in image order permuted within four block rows (each XCD "
                                   "along half a row), later frames longest first by the work counts (rays, shade points) of the frame "
                                   "before -- every frame traces every ray",
                         "first_frame_kernel_ms": round(first["kernel_ms"], 4) if first else None,
                         "off_switch": "RTC_AMD_BLOCK_FEEDBACK=0"},
            "roofline": valu_roofline(pmc, st["kernel_ms"], renderer),
            "contract_hbm_figure": {"algorithmic_bytes": algo_bytes, "achieved_GBps": round(achieved, 2), "peak_GBps": HBM_PEAK_GBS,
                                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                                    "note": "SURVEY.md 8(d) contractual figure: rays x n_objects x 64 B + 48 B/shaded hit + 12 B/pixel "
                                            "over the kernel time. NOT physical traffic (the scene is SGPR-resident and most object "
                                            "tests are culled), so it exceeds the HBM peak; the roof that binds is in `roofline`"},
            "parity_check": verify,
        }
        if dist_on:
            # `value` gathers the Canvas rows (f32: 12 B / pixel; --wire u8: 3) to rank 0 over xGMI: one link per peer, so the
            # step is max(render, rows_of_one_peer / link rate).  Render and transport separately:
            n_parts = world_size + extra_parts
            k_max = float(kern.item())
            line["render_only"] = {"kernel_ms_max_over_ranks": round(k_max, 4),
                                   "value": round(rays / (k_max * 1e-3) / 1e6, 2) if k_max > 0 else None, "unit": "Mrays/s",
                                   "note": "the slowest rank's render kernels per frame (HIP events), no gather: what the band split "
                                           "scales to when the rows stay where they are rendered"}
            if calib and "even_split_ms_per_step" in calib:
                calib["even_split_value_Mrays_per_s"] = round(rays / (calib["even_split_ms_per_step"] * 1e-3) / 1e6, 2)
            line["multi_gpu"] = {"backend": args.backend + (" (RCCL)" if args.backend == "nccl" else ""), "ranks": world_size,
                                 "forced_with_one_rank": bool(args.force_dist and world_size == 1), "wire": args.wire,
                                 "gather": "%s Canvas rows to rank 0 by dist.gather, double-buffered (frame i's gather overlaps frame i+1's render)" % args.wire,
                                 "split": "%d parts; rank 0 renders %d (its rows do not travel), each peer 1" % (n_parts, extra_parts + 1),
                                 "calibration": calib,
                                 "render_kernel_ms_max_over_ranks": round(float(kern.item()), 4),
                                 "gathered_bytes_per_step": int(camera.height * camera.width * (3 if wire_u8 else 12) * (world_size - 1) // n_parts),
                                 "wire_format_gather": wire}
        if world_size == 1 and not args.no_one_shot:
            try:
                line["one_shot"] = one_shot(world, camera, depth)
            except Exception as exc:  # noqa: BLE001 -- informational: never costs the headline line
                line["one_shot"] = {"error": repr(exc)[:300]}
        if args.cpu_seconds > 0 and world_size == 1:
            line["cpu_baseline"] = cpu_baseline(world, camera, depth, args.cpu_seconds)
        elif world_size > 1:
            line["cpu_baseline"] = None  # measured on rank 0 at N=1 only
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def workload_text(args, camera):
    if args.scene == "soft_shadows":
        return "%ssoft_shadows demo scene (10x10 area light, 4 objects, depth 5), %dx%d, hashed jitter seed 0x5EED5EED" % (
            (args.workload_name + ": ") if args.workload_name else "", camera.width, camera.height)
    return "%s%s %dx%d" % ((args.workload_name + ": ") if args.workload_name else "", args.scene, camera.width, camera.height)


def workload_key(scene, width, height):
    return "%s:%dx%d" % (scene, width, height)


def matching_pmc_summary(kernel_id, scene, width, height):
    """The committed rocprofv3 counter summary (profiles/*_pmc.json, written by profiles/summarize.py) that was
    measured on THIS kernel (same code id) and THIS workload -- or None: counts are never quoted for another kernel."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            m = json.load(open(path))
        except Exception:
            continue
        if m.get("kernel_id") == kernel_id and m.get("workload_key") == workload_key(scene, width, height):
            best = (path, m)
    return best


def live_pmc(args, kernel_id):
    """The dominant kernel's counters measured NOW: this script again, a few steps, as a child of `rocprofv3 --pmc <one group>` (never
    combined with a trace domain; the program itself after `--`), once per group -- VALU instructions, then the two HBM-side byte
    counters of MI355X_MICROARCH.md's recipe (separate passes; bytes = (2 FETCH_SIZE + WRITE_SIZE) x 1024).  Medians over the frames'
    dispatches (the one-workgroup warm-up dispatch left out).  None if rocprofv3 is missing, fails, times out, or profiled another
    kernel than the one this process timed: the caller falls back to the committed summary."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):  # (already under a profiler: not twice)
        return None
    child = [sys.executable, os.path.abspath(__file__), "--steps", "4", "--warmup", "2", "--cpu-seconds", "0", "--no-verify", "--no-one-shot", "--no-live-pmc",
             "--scene", args.scene, "--size", str(args.width), "--height", str(args.height_px)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    out = {}
    tmp = tempfile.mkdtemp(prefix="rtc_pmc_", dir="/tmp")
    try:
        for group in (["SQ_INSTS_VALU"], ["FETCH_SIZE"], ["WRITE_SIZE"]):
            d = os.path.join(tmp, group[0])
            # (its own session: if it outlives its two minutes the whole group goes, the profiled program with the profiler's launcher)
            proc = subprocess.Popen([exe, "--pmc"] + group + ["--output-format", "csv", "-d", d, "--"] + child, env=env, cwd="/tmp", stdout=subprocess.PIPE,
                                    stderr=subprocess.DEVNULL, text=True, start_new_session=True)
            try:
                stdout, _ = proc.communicate(timeout=120)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                return None
            r = proc
            lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines or json.loads(lines[-1])["roofline"]["kernel_id"] != kernel_id:
                return None
            if os.environ.get("RTC_BENCH_PMC_DEBUG"):
                print("live pmc child: %s\n%s" % (" ".join(child), lines[-1][:1500]), file=sys.stderr)
            vals = {}
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "render_kernel" in row["Kernel_Name"] and int(row["Grid_Size"]) > 256:
                        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for name in group:
                v = sorted(vals.get(name, []))
                if not v:
                    return None
                out[name] = v[len(v) // 2]
                if os.environ.get("RTC_BENCH_PMC_DEBUG"):
                    print("live pmc %s: %s" % (name, v), file=sys.stderr)
        return {"counters_mean_per_launch": {"SQ_INSTS_VALU": out["SQ_INSTS_VALU"]}, "hbm_bytes_per_launch": (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0,
                "live": True}
    except Exception:  # noqa: BLE001 -- informational: never costs the headline line
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def valu_roofline(pmc, kernel_ms, renderer):
    out = {"bound": "valu", "achieved": None, "peak": round(VALU_PEAK_TOPS, 1), "unit": "Tlane-op/s", "frac": None, "traffic": None,
           "kernel": renderer.kernel_name, "kernel_id": renderer.kernel_id, "kernel_ms": round(kernel_ms, 4), "pmc_source": None}
    if pmc is None or kernel_ms <= 0:
        out["note"] = ("no profiles/*_pmc.json matches this kernel id and workload: instruction count and HBM traffic are not "
                       "quoted from another kernel's profile (run profiles/run_profile.sh + summarize.py)")
        return out
    path, m = pmc
    c = m["counters_mean_per_launch"]
    lane_ops = c["SQ_INSTS_VALU"] * 64.0
    tops = lane_ops / (kernel_ms * 1e-3) / 1e12
    out.update({"achieved": round(tops, 2), "frac": round(tops / VALU_PEAK_TOPS, 4), "traffic": m.get("hbm_bytes_per_launch"),
                "valu_wave_insts_per_launch": c["SQ_INSTS_VALU"],
                "pmc_source": "live: child runs of this command under rocprofv3 --pmc, one counter group each" if m.get("live") else os.path.relpath(path, ROOT),
                "profiled_kernel_ms": round(m.get("kernel_trace", {}).get("avg_ns", 0.0) / 1e6, 4) if not m.get("live") else None,
                "note": "FP32 VALU issue: PMC SQ_INSTS_VALU x 64 lanes per launch / this run's mean HIP-event kernel time, against "
                        "256 CU x 4 SIMD x 32 lanes x 2.4 GHz; traffic = (2 FETCH_SIZE + WRITE_SIZE) x 1024 B per launch (separate "
                        "--pmc passes), to be read against the %.0f MB f32 canvas store" % (renderer.width * renderer.height * 12 / 1e6)})
    return out


def one_shot(world, camera, depth):
    """The drop-in seam as a caller meets it (camera.rs:76): ONE rtc_render_ex call, host buffer out -- scene check,
    kernel launches, transfer.  Median wall ms of 5 calls after a warm-up call (which pays context, compile, allocation)."""
    import numpy as np
    import ray_tracer_challenge_amd as P
    from ray_tracer_challenge_amd import _lib as L
    import ctypes as C
    res = {}
    h, w = camera.height, camera.width

    def timed(fn):
        fn()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return round(sorted(ts)[2], 3)
    buf = np.zeros((h, w, 3), dtype=np.float32)
    res["f32_pageable_ms"] = timed(lambda: camera.render(world, depth, out=buf))
    res["kernel_ms_in_call"] = round(camera.last_stats["kernel_ms"], 4)
    res["launches_in_call"] = camera.last_stats["launches"]
    buf8 = np.zeros((h, w, 3), dtype=np.uint8)
    res["u8_pageable_ms"] = timed(lambda: camera.render(world, depth, quantize=True, out=buf8))
    res["u8_kernel_ms_in_call"] = round(camera.last_stats["kernel_ms"], 4)
    lib, cs = P.lib(), world._c()
    p = lib.rtc_host_alloc(h * w * 12)
    if p:
        st = L.rtc_stats()
        dev = (C.c_int32 * 1)(0)
        for name, q in (("f32_pinned_ms", 0), ("u8_pinned_ms", 1)):
            opts = L.rtc_opts(dev, 1, 0, q, 0)
            res[name] = timed(lambda: L.check(lib.rtc_render_ex(C.byref(cs.scene), C.byref(camera._cam), depth, C.byref(opts),
                                                                C.c_void_p(p), C.byref(st))))
            res[name.replace("_ms", "_kernel_ms_in_call")] = round(float(st.kernel_ms), 4)
        lib.rtc_host_free(p)
    res["note"] = ("rtc_render_ex, %dx%d, whole call: ONE launch whose kernel reports finished chunks of rows, each leaving as it is "
                   "reported; pageable: DMA into pinned staging + threaded copy into the caller's buffer; pinned: DMA straight into an "
                   "rtc_host_alloc buffer; u8: scale_color'd bytes stored by the render kernel itself" % (w, h))
    lib.rtc_render_release()
    return res


def verify_rows(image, world, camera, depth):
    """Before reporting a number: a few full rows of the timed frame against the CPU oracle, bit-exact."""
    import numpy as np

    from tests import helpers as H
    ow, oc = H.oracle_world(world), H.oracle_camera(camera)
    h = camera.height
    cores = min(16, len(os.sched_getaffinity(0)))
    checked = []
    for y in sorted({h // 3, (h * 5) // 8, h - 2}):
        if y < 0 or y >= h:
            continue
        exp, _ = oc.render(ow, depth, threads=cores, rows=(y, y + 1))
        got = image[y].cpu().numpy()
        if not np.array_equal(got == exp[y], np.ones_like(got, dtype=bool)):
            raise SystemExit("parity check FAILED on row %d: refusing to report a throughput number" % y)
        checked.append(y)
    return {"rows_checked_bit_exact_vs_oracle": checked}


if __name__ == "__main__":
    main()
