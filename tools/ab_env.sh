#!/bin/bash
# usage: tools/ab_env.sh "VAR=1 VAR2=x" "VAR=0" ...   one short stage-timed bench run per environment string
for v in "$@"; do
  echo "== $v"
  env $v RPT_STAGE_TIMING=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-workloads ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['stage_ms'], 'launches', r['launches'], 'rays/launch', r['units_per_launch'])"
done
