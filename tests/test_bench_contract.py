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


def test_committed_round2_summaries_carry_their_stamps():
    stamped = [p for p in glob.glob(os.path.join(ROOT, "profiles", "r02*_pmc.json"))]
    assert stamped, "no round-2 summaries committed"
    for p in stamped:
        m = json.load(open(p))
        assert m["kernel_id"].startswith(("spec_", "aot_")) and ":" in m["workload_key"], p
        assert m["counters_mean_per_launch"]["SQ_INSTS_VALU"] > 0 and m["kernel_trace"]["avg_ns"] > 0, p
