#!/bin/bash
# tools/ray_sort_probe.py under rocprofv3 --kernel-trace: the production nearest-hit walk on the same rays in four slot orders (three launches each; the least is printed)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rsp
rocprofv3 --kernel-trace --output-format csv -d /tmp/rsp -- python3 "$ROOT/tools/ray_sort_probe.py" "$@" > /tmp/rsp.log 2>&1
grep "^ORDER\|Error\|error" /tmp/rsp.log
python3 - "$(find /tmp/rsp -name '*kernel_trace.csv' | head -1)" /tmp/rsp.log <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_traverse_nearest" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [l.split("|")[0][6:].strip() for l in open(sys.argv[2]) if l.startswith("ORDER")]
ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
base = None
for k, name in enumerate(names):
    best = min(ms[3 * k: 3 * k + 3])
    base = base or best
    print(f"{name}: walk {best:.2f} ms ({base / best:.2f} x)   [{rows[3 * k]['Kernel_Name'][:60]}]")
PY
