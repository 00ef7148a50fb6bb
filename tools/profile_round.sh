#!/bin/bash
# Produce the judged evidence of a round on the GPU box into gpurun_out/profile/: bench line, rocprofv3 kernel stats,
# FETCH_SIZE / WRITE_SIZE passes (each PMC counter in its own run, --kernel-trace only).  usage: tools/profile_round.sh TAG
set -u
TAG=${1:-r01_final}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_n1.json"
rm -rf /tmp/prof_stats /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/${TAG}_stats_run.log" 2>&1
cp "$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_darkcornell_kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -- python3 "$ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -- python3 "$ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
cp "$(find /tmp/prof_f -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_darkcornell_pmc_FETCH_SIZE.csv"
cp "$(find /tmp/prof_w -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_darkcornell_pmc_WRITE_SIZE.csv"
python3 "$ROOT/tools/traffic_from_pmc.py" "$OUT/${TAG}_darkcornell_pmc_FETCH_SIZE.csv" "$OUT/${TAG}_darkcornell_pmc_WRITE_SIZE.csv" darkcornell k_traverse_nearest "$OUT/traffic_latest.json" \
  "profiles/${TAG}_darkcornell_pmc_FETCH_SIZE.csv, profiles/${TAG}_darkcornell_pmc_WRITE_SIZE.csv (separate --pmc passes, bench.py --steps 2 --warmup 0 --no-cpu-baseline)"
tail -1 "$OUT/${TAG}_stats_run.log" | cut -c1-300
head -6 "$OUT/${TAG}_darkcornell_kernel_stats.csv"
cat "$OUT/${TAG}_bench_n1.json"
