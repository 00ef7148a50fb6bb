#!/bin/bash
# per-dispatch counter values of one kernel in launch order.  usage: tools/pmc_per_dispatch.sh KERNEL_SUBSTRING "COUNTER ..." [bench args]
K=$1; C=$2; shift 2
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_pd
rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_pd -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra-workloads --no-readback --no-parity-check "$@" > /tmp/pd.log 2>&1
python3 - "$K" "$(find /tmp/prof_pd -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
k, path = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    if k in r["Kernel_Name"]:
        rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for d, v in list(rows.items())[-12:]:
    print(d, {a: round(b) for a, b in v.items()})
PY
