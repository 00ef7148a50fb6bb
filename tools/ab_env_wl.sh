#!/bin/bash
# usage: tools/ab_env_wl.sh "workload ..." "ENV=.. ENV=.." ["ENV=.."] ...  one short stage-timed bench run per environment string and workload
WLS=$1; shift
for v in "$@"; do
  for wl in $WLS; do
    env $v timeout 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-extra-workloads --no-readback --no-parity-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', '[$v]', d['value'], d['roofline']['stage_ms'])"
  done
done
