#!/bin/bash
# A/B of two builds on the textured PBRTest workload (two interleaved rounds).  usage: tools/ab_textured.sh base.so
cd ${GRAFT_REPO_ROOT:-.}
for round in 1 2; do
  for lib in "$1" ""; do
    RPT_HIP_LIB=$lib RPT_STAGE_TIMING=1 timeout 300 python bench.py --workload pbrtest_textured --steps 4 --warmup 2 --no-cpu-baseline --no-extra-workloads --no-readback 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pbrtest_textured', '[${lib:-in-tree}]', d['value'], {k: round(v / 4, 3) for k, v in d['roofline']['stage_ms'].items()}, d['parity_check']['bitwise'])"
  done
done
