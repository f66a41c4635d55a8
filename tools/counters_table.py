#!/usr/bin/env python3
"""The counters table of DESIGN.md section 8 from profiles/<tag>_*_pmc.json (documentation tool): python tools/counters_table.py r03"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROWS = [("c3", "C3 4096²", 201), ("c4", "C4 4096²", 201), ("c5", "C5 8192²", 805), ("hexagons", "hexagons 4096×2048", 101),
        ("first_textures", "first_textures 4096×2048", 101), ("reflect_refract", "reflect_refract 4096×2048", 101), ("mesh", "mesh 2048²", 50),
        ("dragons", "here_be_dragons 1000×400", 5)]
print("| scene | ms (kernel trace, mean of 13 launches) | VALU wave-instr | cycles / instr | lanes on | issuing / waiting to issue / waiting on memory | HBM-side MB (canvas) | frac |")
print("|---|---|---|---|---|---|---|---|")
for key, label, canvas in ROWS:
    m = json.load(open(os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (tag, key))))
    c, d = m["counters_mean_per_launch"], m["derived"]
    s = d["wave_cycle_split"]
    print("| %s | %.3f | %.0f M | %.1f | %.0f %% | %.0f / %.0f / %.0f %% | %.0f (%d) | %.2f |" % (
        label, m["kernel_trace"]["avg_ns"] / 1e6, c["SQ_INSTS_VALU"] / 1e6, d["cycles_per_valu_wave_inst"], 100 * d["valu_lane_utilisation"],
        100 * s["SQ_ACTIVE_INST_ANY"], 100 * s["SQ_WAIT_INST_ANY"], 100 * s["SQ_WAIT_ANY"], m["hbm_bytes_per_launch"] / 1e6, canvas,
        d["valu_issue_frac_of_78_6_Tlaneops"]))
