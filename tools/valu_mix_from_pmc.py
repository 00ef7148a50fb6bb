#!/usr/bin/env python3
"""Adds the measured instruction-CLASS mix of every stage kernel to a traffic_<workload>.json, and from it the SIMD issue cycles one of its
wave64 VALU instructions needs on gfx950 — what bench.py's `issue_frac` multiplies the instruction counts with.

Classes (two --pmc passes, tools/profile_workload.sh `sq3` / `sq4`):
  fast    SQ_INSTS_VALU_{ADD,MUL,FMA}_F32                     2.1 SIMD cycles per wave64 instruction (tools/microbench/valu_rates.hip)
  trans   SQ_INSTS_VALU_TRANS_F32 (v_rcp / v_sqrt / v_rsq)     7
  slow    everything else: SQ_INSTS_VALU_{ADD,MUL,FMA}_F64, _INT32, _INT64, _CVT and what no class counter sees (v_min / v_max / v_cmp /
          v_cndmask / v_mov / v_readlane ...): 3.45 — but a mix hides up to HALF as many slow instructions as it has fast ones behind
          them (profiles/r04_valu_pipes.txt: 48 v_fma + 24 v_min issue in the time of 72 v_fma, in any order)
usage: valu_mix_from_pmc.py <tag>_<wl>_pmc_sq3.txt <tag>_<wl>_pmc_sq4.txt traffic_<wl>.json"""
import ast
import json
import re
import sys

STAGES = (("k_traverse_nearest", "traverse"), ("k_shade", "shade"), ("k_traverse_shadow", "shadow"), ("k_sky", "sky"),
          ("k_generate_first", "generate"), ("k_complete", "complete"), ("k_shadow_resolve", "shadow_resolve"))
FAST, TRANS, SLOW = 2.1, 7.0, 3.45


def read(path):
    acc = {}
    for line in open(path):
        m = re.match(r"^(.*?) (\{.*\}) launches (\d+)\s*$", line)
        if not m:
            continue
        name, vals, n = m.group(1), ast.literal_eval(m.group(2)), int(m.group(3))
        for sub, stage in STAGES:
            if sub in name:
                a = acc.setdefault(stage, {})
                for k, v in vals.items():
                    a[k] = a.get(k, 0.0) + n * v
                a["launches"] = a.get("launches", 0) + n
    return acc


f32, f64, out = sys.argv[1:4]
a32, a64 = read(f32), read(f64)
j = json.load(open(out))
for stage, a in a32.items():
    if stage not in j["stages"] or "valu" not in j["stages"][stage] or a.get("SQ_INSTS_VALU", 0) <= 0:
        continue
    total = a["SQ_INSTS_VALU"]
    b = a64.get(stage, {})
    scale = total / b["SQ_INSTS_VALU"] if b.get("SQ_INSTS_VALU", 0) > 0 else 0.0       # (the two passes are separate runs of the same launches)
    fast = a.get("SQ_INSTS_VALU_ADD_F32", 0) + a.get("SQ_INSTS_VALU_MUL_F32", 0) + a.get("SQ_INSTS_VALU_FMA_F32", 0)
    trans = a.get("SQ_INSTS_VALU_TRANS_F32", 0)
    f64n = scale * (b.get("SQ_INSTS_VALU_ADD_F64", 0) + b.get("SQ_INSTS_VALU_MUL_F64", 0) + b.get("SQ_INSTS_VALU_FMA_F64", 0))
    int_n = a.get("SQ_INSTS_VALU_INT32", 0) + scale * b.get("SQ_INSTS_VALU_INT64", 0)
    cvt = a.get("SQ_INSTS_VALU_CVT", 0)
    slow = max(total - fast - trans, 0.0)
    hidden = min(slow, fast / 2.0)
    cycles = (fast + hidden) * FAST + (slow - hidden) * SLOW + trans * TRANS
    v = j["stages"][stage]["valu"]
    v["class_mix"] = {"fma_mul_add_f32": round(fast / total, 4), "trans_f32": round(trans / total, 4), "f64": round(f64n / total, 4),
                      "int": round(int_n / total, 4), "cvt": round(cvt / total, 4),
                      "unclassified_min_max_cmp_select_mov": round(max(total - fast - trans - f64n - int_n - cvt, 0.0) / total, 4)}
    v["issue_cycles_per_wave_instruction"] = round(cycles / total, 3)
j["valu_mix_source"] = ("SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32, _INT32, _CVT (" + f32.split("/")[-1] + ") and _{ADD,MUL,FMA}_F64, _INT64 (" + f64.split("/")[-1] +
                        "): fast class 2.1, trans 7, the rest 3.45 SIMD cycles with up to fast / 2 of it hidden (profiles/r04_valu_pipes.txt)")
json.dump(j, open(out, "w"), indent=1)
print({s: (v.get("valu", {}).get("class_mix"), v.get("valu", {}).get("issue_cycles_per_wave_instruction")) for s, v in j["stages"].items()})
