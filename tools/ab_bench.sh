#!/bin/bash
# A/B of bench workloads under environment knobs.  usage: tools/ab_bench.sh "workload args" "ENV=.. ENV=.." ["ENV=.." ...]
# prints value / ms_per_step / stage_ms per variant.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=$1; shift
for envs in "$@"; do
  out=$(env $envs python3 "$ROOT/bench.py" --no-cpu-baseline --no-extra-workloads --no-readback $WL 2>/dev/null | tail -1)
  python3 - "$envs" "$WL" "$out" <<'PY'
import json, sys
envs, wl, out = sys.argv[1:4]
try:
    j = json.loads(out)
    print(f"{wl:40s} [{envs}] {j['value']:9.1f} Mrays/s {j['ms_per_step']:9.3f} ms/step stages {j['roofline']['stage_ms']}")
except Exception as e:
    print(wl, envs, "FAILED", out[:200])
PY
done
