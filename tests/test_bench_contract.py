"""bench.py's measurement block (CPU): counters are quoted only for the kernel that ran.

`roofline.achieved / frac / traffic` come from a committed rocprofv3 summary (profiles/*_pmc.json); a summary measured on
another kernel (a different rtc_ctx_kernel_id) or another workload must never be used -- the block then carries nulls
and says why.  The committed summaries themselves must carry the stamps bench.py matches on.
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class _FakeRenderer:
    kernel_name, kernel_id, width, height = "render_kernel_spec[test]", "spec_0123456789abcdef", 4096, 4096


def test_roofline_block_without_a_matching_summary_has_nulls():
    r = bench.valu_roofline(None, 0.9, _FakeRenderer())
    assert r["bound"] == "valu" and r["achieved"] is None and r["frac"] is None and r["traffic"] is None
    assert r["kernel_id"] == _FakeRenderer.kernel_id and "no profiles/*_pmc.json matches" in r["note"]
    assert abs(r["peak"] - 78.6) < 0.05


def test_roofline_block_from_a_matching_summary():
    m = {"counters_mean_per_launch": {"SQ_INSTS_VALU": 680.0e6}, "hbm_bytes_per_launch": 234.0e6, "kernel_trace": {"avg_ns": 922000.0}}
    r = bench.valu_roofline((os.path.join(ROOT, "profiles", "x_pmc.json"), m), 0.823, _FakeRenderer())
    lane_ops_per_s = 680.0e6 * 64 / 0.823e-3
    assert abs(r["achieved"] - lane_ops_per_s / 1e12) < 0.01
    assert abs(r["frac"] - lane_ops_per_s / 1e12 / bench.VALU_PEAK_TOPS) < 1e-3 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] == 234.0e6 and r["pmc_source"] == os.path.join("profiles", "x_pmc.json")


def test_matching_is_by_kernel_id_and_workload(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    base = {"counters_mean_per_launch": {"SQ_INSTS_VALU": 1.0}, "workload_key": bench.workload_key("soft_shadows", 4096, 4096)}
    json.dump(dict(base, kernel_id="spec_aaaa"), open(tmp_path / "profiles" / "a_pmc.json", "w"))
    json.dump(dict(base, kernel_id="spec_bbbb", workload_key=bench.workload_key("mesh", 2048, 2048)), open(tmp_path / "profiles" / "b_pmc.json", "w"))
    json.dump({"counters_mean_per_launch": {}}, open(tmp_path / "profiles" / "old_pmc.json", "w"))   # a round-1 summary: no stamps
    assert bench.matching_pmc_summary("spec_aaaa", "soft_shadows", 4096, 4096)[0].endswith("a_pmc.json")
    assert bench.matching_pmc_summary("spec_aaaa", "soft_shadows", 1000, 400) is None     # same kernel, other workload
    assert bench.matching_pmc_summary("spec_bbbb", "soft_shadows", 4096, 4096) is None     # other kernel
    assert bench.matching_pmc_summary("spec_cccc", "mesh", 2048, 2048) is None


def test_committed_summaries_carry_their_stamps():
    stamped = [p for p in glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]*_pmc.json"))]  # (this round's; earlier rounds' are under history/)
    assert stamped, "no stamped summaries committed"
    for p in stamped:
        m = json.load(open(p))
        assert m["kernel_id"].startswith(("spec_", "aot_")) and ":" in m["workload_key"], p
        assert m["counters_mean_per_launch"]["SQ_INSTS_VALU"] > 0 and m["kernel_trace"]["avg_ns"] > 0, p


# ---- `python bench.py --gpus N` with no launcher around it starts its own ranks ------------------------------------
def _run_bench(args, env_extra=None, timeout=600):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_workload_selector_names_the_baseline_configurations():
    assert bench.WORKLOADS["C3"] == ("soft_shadows", 4096, 4096) and bench.WORKLOADS["C5"] == ("sphere_grid", 8192, 8192)
    a = bench.parse([])
    assert (a.scene, a.width, a.height_px, a.workload_name) == ("soft_shadows", 4096, 4096, "C3")  # the metric configuration
    a = bench.parse(["--workload", "C5", "--gpus", "8"])
    assert (a.scene, a.width, a.height_px, a.workload_name, a.gpus) == ("sphere_grid", 8192, 8192, "C5", 8)
    a = bench.parse(["--scene", "mesh", "--size", "2048"])
    assert (a.scene, a.width, a.height_px, a.workload_name) == ("mesh", 2048, 2048, None)
    a = bench.parse(["--size", "1000", "--height", "400"])
    assert (a.scene, a.width, a.height_px, a.workload_name) == ("soft_shadows", 1000, 400, "C1")


def test_self_launching_parent_touches_no_gpu_api_and_reports_failed_ranks():
    """The parent of a self-launched N > 1 run must not initialise the GPU (a process that has may not hand its work to
    children): it imports neither torch nor the render library.  Its ranks are real `bench.py` processes; here a stand-in
    child command (RANK-dependent exit codes) shows that every rank gets its rendezvous environment, that the parent
    exits non-zero when any rank does, and that it ends the others."""
    import subprocess
    import textwrap
    code = textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r)
        import bench
        seen = []
        import subprocess
        class P(subprocess.Popen):
            def __init__(self, cmd, env=None, **kw):
                seen.append({k: env[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')})
                assert cmd[1].endswith('bench.py') and cmd[2:] == ['--gpus', '3', '--steps', '2']
                # the stand-in rank: rank 1 fails at once, the others would run for a minute
                body = 'import os,sys,time; r=int(os.environ["RANK"]); sys.exit(7) if r == 1 else time.sleep(60)'
                super().__init__([sys.executable, '-c', body], env=env, **kw)
        subprocess.Popen = P
        try:
            bench.main(['--gpus', '3', '--steps', '2'])
        except SystemExit as e:
            rc = e.code
        print(json.dumps({'rc': rc, 'seen': seen, 'torch': 'torch' in sys.modules,
                          'rtc': any(m.startswith('ray_tracer_challenge_amd') for m in sys.modules)}))
        """ % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["rc"] == 7 and time.time() - t0 < 50, out  # the failing rank's code; the sleeping ranks were ended, not waited for
    assert out["torch"] is False and out["rtc"] is False   # no GPU API anywhere near the parent
    assert [s["RANK"] for s in out["seen"]] == ["0", "1", "2"] and {s["WORLD_SIZE"] for s in out["seen"]} == {"3"}
    assert {s["MASTER_ADDR"] for s in out["seen"]} == {"127.0.0.1"} and len({s["MASTER_PORT"] for s in out["seen"]}) == 1
    assert {s["HSA_ENABLE_IPC_MODE_LEGACY"] for s in out["seen"]} == {"0"}
    assert "rank 1 exited with 7" in r.stderr


def test_self_launched_ranks_fail_loudly_without_a_gpu():
    """On a box without a GPU the real ranks refuse to run (no CPU fallback) and the parent's exit code says so."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "64"], timeout=300)
    assert r.returncode != 0 and "needs a GPU" in r.stderr and r.stdout.strip() == ""


def test_a_launcher_provided_world_size_must_match():
    r = _run_bench(["--gpus", "2", "--steps", "1"], env_extra={"WORLD_SIZE": "4", "RANK": "0"}, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_live_counters_are_tagged_and_never_collected_under_a_profiler(monkeypatch):
    """Counters bench.py measured itself (child runs under rocprofv3 --pmc) are tagged `live`; a bench.py that is itself
    running under rocprofv3 must not start profilers of its own."""
    m = {"counters_mean_per_launch": {"SQ_INSTS_VALU": 387.4e6}, "hbm_bytes_per_launch": 235.0e6, "live": True}
    r = bench.valu_roofline(("live", m), 0.56, _FakeRenderer())
    assert r["pmc_source"].startswith("live") and r["profiled_kernel_ms"] is None and r["traffic"] == 235.0e6
    assert abs(r["frac"] - 387.4e6 * 64 / 0.56e-3 / 1e12 / bench.VALU_PEAK_TOPS) < 1e-3
    monkeypatch.setenv("ROCPROF_COUNTER_COLLECTION", "1")
    assert bench.live_pmc(bench.parse(["--steps", "1"]), "spec_x.y") is None


def test_kernel_ids_of_the_committed_summaries_name_the_code_object():
    """Since round 4 a scene kernel's id ends in the checksum of its code object (two compilers gave two binaries for one source):
    the summaries of the round carry that form."""
    import re
    for p in glob.glob(os.path.join(ROOT, "profiles", "r04_*_pmc.json")):
        kid = json.load(open(p))["kernel_id"]
        assert re.fullmatch(r"spec_[0-9a-f]{16}\.[0-9a-f]{8}", kid), (p, kid)
