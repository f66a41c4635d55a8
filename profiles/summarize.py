#!/usr/bin/env python3
"""Turns a gpurun_out/prof_<tag>/ directory (profiles/run_profile.sh) into the committed summaries:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 5`
   profiles/<tag>_pmc.json           per-launch PMC means for render_kernel + derived figures
   profiles/hbm_traffic.json         HBM bytes per launch (read by bench.py for roofline.traffic)
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
m = {k: sum(v) / len(v) for k, v in agg.items()}
out = {"tag": tag, "launch": meta, "counters_mean_per_launch": m}
for r in csv.DictReader(open(stats)):
    if "render_kernel" in r["Name"]:
        out["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                               "max_ns": float(r["MaxNs"])}
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
    hbm = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    out["hbm_bytes_per_launch"] = hbm
    json.dump({"workload": "C3 soft_shadows 4096x4096", "hbm_bytes_per_launch": hbm,
               "fetch_size_kb": m["FETCH_SIZE"], "write_size_kb": m["WRITE_SIZE"], "source": "profiles/%s_pmc.json" % tag,
               "valu_wave_insts_per_launch": m.get("SQ_INSTS_VALU"), "kernel": meta.get("kernel"),
               "method": "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes"},
              open(os.path.join(here, "hbm_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(here, "%s_pmc.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1))
