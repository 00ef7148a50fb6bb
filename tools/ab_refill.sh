#!/bin/bash
# interleaved A/B of refill cadences of the streamed global-memory walks (two rounds)
cd $GRAFT_REPO_ROOT
V=rust-path-tracer_amd/lib/variants
for round in 1 2; do
  for lib in "" $V/refill8.so $V/trips4_refill8.so $V/trips4_refill16.so; do
    for wl in veachmis pbrtest; do
      RPT_HIP_LIB=$lib RPT_STAGE_TIMING=1 timeout 300 python bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --no-extra-workloads --no-readback --no-parity-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', '[${lib:-in-tree: trips 8 / refill 24}]', d['value'], {k: round(v / 4, 3) for k, v in d['roofline']['stage_ms'].items()})"
    done
  done
done
