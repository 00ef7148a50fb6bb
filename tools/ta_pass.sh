#!/bin/bash
# Texture-address-path counters of the walks (what bounds the global-memory walks: DESIGN.md 4): per-kernel per-launch averages into
# gpurun_out/profile/${TAG}_${WL}_pmc_ta.txt.   usage: tools/ta_pass.sh TAG workload [workload ...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out/profile"
for WL in "$@"; do
  BENCH_ARGS="--workload $WL --no-readback --no-parity-check" bash "$ROOT/tools/pmc_pass.sh" \
    "TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum" \
    "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
    "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" 2>&1 | grep -E "k_traverse|k_shade" > "$ROOT/gpurun_out/profile/${TAG}_${WL}_pmc_ta.txt"
  echo "== $WL"; cut -c1-400 "$ROOT/gpurun_out/profile/${TAG}_${WL}_pmc_ta.txt"
done
