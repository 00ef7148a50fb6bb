#!/usr/bin/env python3
"""HBM bytes per launch of every stage kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

usage: traffic_from_pmc.py FETCH.csv WRITE.csv workload out.json [source note]
Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are KiB;
FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads on gfx950 and is doubled; WRITE_SIZE is taken as is.
The file records the fingerprint of the kernel sources it was measured on (tools/source_fingerprint.py); bench.py
only reports `roofline.traffic` from a file whose fingerprint equals that of the build it runs.
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_fingerprint import fingerprint  # noqa: E402

STAGES = (("k_traverse_nearest", "traverse"), ("k_shade", "shade"), ("k_traverse_shadow", "shadow"), ("k_sky", "sky"),
          ("k_generate_first", "generate"), ("k_complete", "complete"), ("k_shadow_resolve", "shadow_resolve"))


def per_launch_kb(path, kernel_sub, counter):
    tot = 0.0
    n = 0
    for r in csv.DictReader(open(path)):
        if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot / max(n, 1), n


if __name__ == "__main__":
    fetch, write, workload, out = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    stages = {}
    for ksub, stage in STAGES:
        f_kb, nf = per_launch_kb(fetch, ksub, "FETCH_SIZE")
        w_kb, nw = per_launch_kb(write, ksub, "WRITE_SIZE")
        if nf == 0 and nw == 0:
            continue
        stages[stage] = {"hbm_bytes_per_launch": int(round((2.0 * f_kb + w_kb) * 1024)),
                         "fetch_size_kb_per_launch": round(f_kb, 1), "write_size_kb_per_launch": round(w_kb, 1),
                         "launches_profiled": [nf, nw], "kernel_matched": ksub}
    json.dump({
        "workload": workload, "stages": stages, "source_fingerprint": fingerprint(),
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled "
                      "(MI355X_MICROARCH.md HBM section); WRITE_SIZE taken as is",
        "source": note,
    }, open(out, "w"), indent=1)
    print(open(out).read())
