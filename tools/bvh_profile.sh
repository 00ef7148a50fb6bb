#!/bin/bash
# Per-launch timing of the GPU BVH build of the 1 M-triangle stand-in (rocprofv3 kernel trace).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bvhprof
rocprofv3 --kernel-trace --output-format csv -d /tmp/bvhprof -- python3 "$ROOT/tools/bvh_profile_target.py" $1 > /dev/null 2>&1
f=$(find /tmp/bvhprof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "bvb" in r["Kernel_Name"]]
gk = [k for k in rows[0].keys() if "Grid" in k][:1] + [k for k in rows[0].keys() if "Workgroup" in k][:1]
rows = rows[[i for i, r in enumerate(rows) if "k_bvb_init" in r["Kernel_Name"]][1]:]            # (after the module-load build of one triangle)
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    if "level" in r["Kernel_Name"] or "small" in r["Kernel_Name"] or "tiny" in r["Kernel_Name"] or r["Kernel_Name"].startswith("k_bvb_team("):
        print(r["Kernel_Name"][:28], [r[k] for k in gk], "at", round((int(r["Start_Timestamp"]) - t0) / 1e6, 2), "ms:", round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3), "ms")
print(len(rows), "launches, kernel time", round(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6, 2), "ms, first to last", round((int(rows[-1]["End_Timestamp"]) - t0) / 1e6, 2), "ms")
PY
