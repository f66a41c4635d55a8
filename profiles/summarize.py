#!/usr/bin/env python3
"""Turns a gpurun_out/prof_<tag>/ directory (profiles/run_profile.sh) into the committed summaries:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 5 [...]`
   profiles/<tag>_pmc.json           per-launch PMC means for render_kernel + derived figures, stamped with the
                                     kernel id (rtc_ctx_kernel_id) and workload key bench.py printed in that very run:
                                     bench.py quotes instruction counts / HBM traffic only from a summary whose stamp
                                     matches the kernel it has just timed.
HBM bytes follow MI355X_MICROARCH.md: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE under-reports
wide reads by 2x on gfx950; WRITE_SIZE is taken as is (our stores are 4-byte, uncalibrated: see README)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = "gpurun_out/prof_%s" % tag
here = os.path.dirname(os.path.abspath(__file__))
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(here, "%s_kernel_stats.csv" % tag))
agg, meta = collections.defaultdict(list), {}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {"kernel": r["Kernel_Name"], "grid": r["Grid_Size"], "workgroup": r["Workgroup_Size"],
                    "vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"], "scratch_bytes_per_lane": r["Scratch_Size"]}
def typical(v):  # the median: the first dispatch of a run (module load, cold caches) can be far off and must not drag a mean
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


m = {k: typical(v) for k, v in agg.items()}
out = {"tag": tag, "launch": meta, "counters_mean_per_launch": m, "counters_are": "medians over the profiled launches"}
for r in csv.DictReader(open(stats)):
    if "render_kernel" in r["Name"]:
        out["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                               "max_ns": float(r["MaxNs"])}
# The one-workgroup warm-up launch (rtc_device.hip ctx_render_slot: a kernel's first launch on a queue, a few microseconds over zero
# rows) is a render_kernel dispatch too: the per-dispatch trace says which rows are frames.
traces = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
if traces:
    frames = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(traces[0]))
              if "render_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) > 256]
    if frames and "kernel_trace" in out:
        out["kernel_trace"].update({"calls_incl_warm_up": out["kernel_trace"]["calls"], "calls": len(frames), "avg_ns": sum(frames) / len(frames),
                                    "min_ns": min(frames), "max_ns": max(frames), "median_ns": sorted(frames)[len(frames) // 2],
                                    "note": "frames only: the one-workgroup warm-up dispatch is left out (its row is in the committed kernel_stats.csv)"})
if "GRBM_GUI_ACTIVE" in m and "SQ_INSTS_VALU" in m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    out["derived"] = {
        "shader_clock_GHz": cyc / out["kernel_trace"]["avg_ns"],
        "valu_wave_insts_per_simd_cycle": m["SQ_INSTS_VALU"] / (cyc * 256 * 4),
        "cycles_per_valu_wave_inst": cyc * 256 * 4 / m["SQ_INSTS_VALU"],
        "valu_lane_utilisation": m.get("SQ_THREAD_CYCLES_VALU", 0) / (m.get("SQ_ACTIVE_INST_VALU", 1) * 64),
        "wave_cycle_split": {k: m[k] / m["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")
                             if k in m},
    }
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    out["hbm_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    out["hbm_method"] = "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes"
# the stamp: what bench.py said it ran, in the traced run and in every counter run (they must agree)
stamps = set()
for log in [os.path.join(src, "trace.log")] + glob.glob(os.path.join(src, "pmc_*.log")) + glob.glob(os.path.join(src, "precompile.log")):
    for line in open(log, errors="replace"):
        if line.startswith("{") and '"roofline"' in line:
            b = json.loads(line)
            stamps.add((b["roofline"]["kernel_id"], b["config"]["workload_key"], b["roofline"]["kernel"]))
            if log.endswith("trace.log"):
                out["bench_line_of_the_traced_run"] = {k: b[k] for k in ("value", "unit", "ms_per_step", "mpixels_per_s") if k in b}
                out["rays_per_frame"] = b["config"]["rays_per_frame"]
                out["shadow_rays_resolved_by_light_cone_cull"] = b["config"]["shadow_rays_resolved_by_light_cone_cull"]
assert len(stamps) == 1, "the profiled runs disagree about the kernel: %r" % (stamps,)
out["kernel_id"], out["workload_key"], out["kernel_name"] = stamps.pop()
if "derived" in out and out.get("rays_per_frame"):
    tested = out["rays_per_frame"] - out["shadow_rays_resolved_by_light_cone_cull"]
    out["derived"]["valu_lane_ops_per_ray"] = m["SQ_INSTS_VALU"] * 64.0 / out["rays_per_frame"]
    out["derived"]["valu_lane_ops_per_tested_ray"] = m["SQ_INSTS_VALU"] * 64.0 / max(tested, 1)
    out["derived"]["valu_issue_frac_of_78_6_Tlaneops"] = m["SQ_INSTS_VALU"] * 64.0 / (out["kernel_trace"]["avg_ns"] * 1e-9) / 78.6432e12
json.dump(out, open(os.path.join(here, "%s_pmc.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1))
