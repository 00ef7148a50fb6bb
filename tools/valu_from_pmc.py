#!/usr/bin/env python3
"""Adds the VALU issue figures of the SQ pass to a traffic_<workload>.json: per stage, wave-instructions per launch and
lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU), launch-weighted over the kernel variants of the stage.
usage: valu_from_pmc.py <tag>_<workload>_pmc_sq.txt traffic_<workload>.json   (the .txt is tools/pmc_summary.py's output)"""
import ast
import json
import re
import sys

STAGES = (("k_traverse_nearest", "traverse"), ("k_shade", "shade"), ("k_traverse_shadow", "shadow"), ("k_sky", "sky"),
          ("k_generate_first", "generate"), ("k_complete", "complete"), ("k_shadow_resolve", "shadow_resolve"))
sq, out = sys.argv[1:3]
acc = {}
for line in open(sq):
    m = re.match(r"^(.*?) (\{.*\}) launches (\d+)\s*$", line)
    if not m:
        continue
    name, vals, n = m.group(1), ast.literal_eval(m.group(2)), int(m.group(3))
    for sub, stage in STAGES:
        if sub in name and "SQ_ACTIVE_INST_VALU" in vals:
            a = acc.setdefault(stage, {"launches": 0, "insts": 0.0, "active": 0.0, "threads": 0.0})
            a["launches"] += n
            a["insts"] += n * vals["SQ_INSTS_VALU"]
            a["active"] += n * vals["SQ_ACTIVE_INST_VALU"]
            a["threads"] += n * vals["SQ_THREAD_CYCLES_VALU"]
j = json.load(open(out))
for stage, a in acc.items():
    if stage in j["stages"] and a["active"] > 0:
        j["stages"][stage]["valu"] = {"wave_instructions_per_launch": int(round(a["insts"] / a["launches"])),
                                      "lane_utilisation": round(a["threads"] / (64.0 * a["active"]), 4),
                                      "launches_profiled": a["launches"]}
j["valu_source"] = "SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_THREAD_CYCLES_VALU of the SQ pass of the same script run (" + sq.split("/")[-1] + ")"
json.dump(j, open(out, "w"), indent=1)
print({s: v.get("valu") for s, v in j["stages"].items()})
