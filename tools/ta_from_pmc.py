#!/usr/bin/env python3
"""Adds the texture-address unit's busy share per stage to a traffic_<workload>.json: TA_TA_BUSY_sum / TCP_GATE_EN1_sum of the stage's kernels
(launch-weighted), from the `ta` pass of tools/profile_workload.sh.  The streamed global-memory walks are bound by that unit, not by HBM.
usage: ta_from_pmc.py <tag>_<workload>_pmc_ta.txt traffic_<workload>.json"""
import ast
import json
import re
import sys

STAGES = (("k_traverse_nearest", "traverse"), ("k_shade", "shade"), ("k_traverse_shadow", "shadow"), ("k_sky", "sky"),
          ("k_generate_first", "generate"), ("k_complete", "complete"), ("k_shadow_resolve", "shadow_resolve"))
src, out = sys.argv[1:3]
acc = {}
for line in open(src):
    m = re.match(r"^(.*?) (\{.*\}) launches (\d+)\s*$", line)
    if not m:
        continue
    name, vals, n = m.group(1), ast.literal_eval(m.group(2)), int(m.group(3))
    if "TA_TA_BUSY_sum" not in vals or "TCP_GATE_EN1_sum" not in vals:
        continue
    for sub, stage in STAGES:
        if sub in name:
            a = acc.setdefault(stage, {"busy": 0.0, "clk": 0.0, "loads": 0.0, "launches": 0})
            a["busy"] += n * vals["TA_TA_BUSY_sum"]
            a["clk"] += n * vals["TCP_GATE_EN1_sum"]
            a["loads"] += n * vals.get("TA_FLAT_READ_WAVEFRONTS_sum", 0.0)
            a["launches"] += n
j = json.load(open(out))
for stage, a in acc.items():
    if stage in j["stages"] and a["clk"] > 0:
        j["stages"][stage]["ta"] = {"busy_frac": round(a["busy"] / a["clk"], 4),
                                    "read_wave_instructions_per_launch": int(round(a["loads"] / a["launches"])),
                                    "busy_cycles_per_read_wave_instruction": round(a["busy"] / a["loads"], 2) if a["loads"] else None}
j["ta_source"] = "TA_TA_BUSY_sum / TCP_GATE_EN1_sum (texture-address unit busy cycles over the L1's clocks, summed over the CUs) of the ta pass (" + src.split("/")[-1] + ")"
json.dump(j, open(out, "w"), indent=1)
print({s: v.get("ta") for s, v in j["stages"].items()})
