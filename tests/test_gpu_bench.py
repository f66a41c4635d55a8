"""bench.py itself on the GPU box: the one-GPU line, and the N > 1 path started the way the driver starts it --
`python bench.py --gpus N` with no launcher -- rehearsed with two gloo ranks sharing the one GPU (RCCL needs a device per
rank; everything else -- the self-launch, the band split, the pipelined gather, the calibration, the parity rows -- is the
code an 8-GPU node runs)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_one_gpu_line_and_self_launched_two_rank_line_agree():
    one = _bench(["--size", "512", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--no-one-shot", "--no-live-pmc"])
    assert one["n_gpus"] == 1 and one["unit"] == "Mrays/s" and one["scaling"] == "strong" and one["dtype"] == "f32"
    assert one["parity_check"]["rows_checked_bit_exact_vs_oracle"]
    two = _bench(["--gpus", "2", "--backend", "gloo", "--size", "512", "--steps", "3", "--warmup", "1"])
    assert two["n_gpus"] == 2 and two["steps"] == 3 and two["value"] > 0 and two["ms_per_step"] > 0
    # the job is one image whatever N is: same rays, same shaded hits, same pixels
    for key in ("rays_per_frame", "shaded_hits_per_frame", "pixels_per_frame"):
        assert two["config"][key] == one["config"][key], key
    # (how many of those rays the light-cone cull answered is decided per WAVE -- a vote over its 64 lanes -- and which pixels
    # share a wave differs between the two runs: their block lists are cut by their own frames' wave times)
    a, b = one["config"]["shadow_rays_resolved_by_light_cone_cull"], two["config"]["shadow_rays_resolved_by_light_cone_cull"]
    assert abs(a - b) <= 0.01 * a, (a, b)
    assert two["parity_check"]["rows_checked_bit_exact_vs_oracle"]  # rows of the GATHERED frame against the oracle
    mg = two["multi_gpu"]
    assert mg["calibration"]["render_ms_even_share"] > 0 and mg["calibration"]["gather_ms_even_share_unoverlapped"] > 0
    assert mg["wire_format_gather"]["equals_quantised_f32_frame"] is True
    assert two["cpu_baseline"] is None and "one_shot" not in two


def test_workload_selector_runs_c5_by_name():
    line = _bench(["--workload", "C5", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--no-one-shot", "--no-live-pmc"])
    assert line["config"]["workload"].startswith("C5: sphere_grid 8192x8192") and line["config"]["pixels_per_frame"] == 8191 * 8191
    assert line["parity_check"]["rows_checked_bit_exact_vs_oracle"]


def test_the_drivers_launcher_form_still_works():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`:
    the ranks come from the launcher (WORLD_SIZE / RANK in the environment), bench.py must not start its own."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size", "384",
                        "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["parity_check"]["rows_checked_bit_exact_vs_oracle"] and line["multi_gpu"]["calibration"]


def test_the_rccl_branch_runs_with_one_rank():
    """`--force-dist` with the default backend: init_process_group("nccl", device_id=...) -- RCCL loads and a communicator is
    created --, the band gather through dist.gather on the communication stream, work.wait()'s stream hand-off, the
    calibration and the other wire format, all with ONE rank on the one GPU this box has.  The gathered frame's rows are
    compared with the oracle like any other line's."""
    line = _bench(["--force-dist", "--size", "512", "--steps", "3", "--warmup", "2", "--cpu-seconds", "0", "--no-one-shot", "--no-live-pmc"])
    assert line["n_gpus"] == 1 and line["value"] > 0
    mg = line["multi_gpu"]
    assert mg["backend"] == "nccl (RCCL)" and mg["ranks"] == 1 and mg["forced_with_one_rank"] is True and mg["wire"] == "f32"
    assert mg["calibration"]["gather_ms_even_share_unoverlapped"] > 0
    assert mg["wire_format_gather"]["equals_quantised_f32_frame"] is True
    assert line["render_only"]["value"] >= line["value"] * 0.5
    assert line["parity_check"]["rows_checked_bit_exact_vs_oracle"]  # rows of the frame dist.gather delivered
    # ... and with the bytes Canvas::to_ppm prints as the measured wire format
    u8 = _bench(["--force-dist", "--wire", "u8", "--size", "512", "--steps", "3", "--warmup", "2", "--cpu-seconds", "0", "--no-one-shot", "--no-live-pmc"])
    assert u8["multi_gpu"]["wire"] == "u8" and u8["multi_gpu"]["wire_format_gather"]["equals_quantised_f32_frame"] is True
    assert u8["parity_check"]["rows_checked_bit_exact_vs_oracle"]
    assert u8["config"]["rays_per_frame"] == line["config"]["rays_per_frame"]


def test_the_roofline_counters_are_collected_live():
    """Without --no-live-pmc the one-GPU line measures its own counters (child runs under rocprofv3 --pmc, one group each): the
    VALU instruction count and the HBM-side bytes belong to the kernel that was timed, whatever profiles/ holds."""
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("no rocprofv3 on this box")
    line = _bench(["--size", "1024", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--no-one-shot"])
    roof = line["roofline"]
    assert roof["pmc_source"].startswith("live"), roof
    pixels = 1023 * 1023
    assert roof["valu_wave_insts_per_launch"] > pixels / 64 * 100  # (hundreds of VALU instructions per pixel at the least)
    assert roof["traffic"] >= 0.5 * pixels * 12 and roof["traffic"] < 40 * pixels * 12  # the canvas store, and not absurdly more
    assert 0 < roof["frac"] < 1
