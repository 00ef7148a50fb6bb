#!/bin/bash
# One rocprofv3 PMC pass per counter group over a short bench run; prints per-kernel averages.
# usage (on the GPU box): tools/pmc_pass.sh "SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS" ...
# Counters are collected in their own runs (--pmc + --kernel-trace only).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for group in "$@"; do
  i=$((i+1))
  d=$OUT/pass$i
  rm -rf "$d"
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$d" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra-workloads ${BENCH_ARGS:-} > "$d.log" 2>&1
  f=$(find "$d" -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / n[(k, c)], 1) for c, v in acc[k].items()}, "launches", max(n[(k, c)] for c in acc[k]))
PY
done
