#!/bin/bash
# A/B of differently built librpt_hip.so files on bench workloads.  usage: tools/ab_libs.sh "workload [workload ...]" lib.so [lib.so ...]   ("" = the in-tree build)
WLS=$1; shift
for lib in "$@"; do
  for wl in $WLS; do
    RPT_HIP_LIB=$lib timeout 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-extra-workloads --no-readback --no-parity-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', '[$lib]', d['value'], d['roofline']['stage_ms'])"
  done
done
