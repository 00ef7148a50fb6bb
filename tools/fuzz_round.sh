#!/bin/bash
# The randomised differential runs of a round, tallies into gpurun_out/fuzz_$1.txt.  usage: tools/fuzz_round.sh TAG
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/fuzz_$TAG.txt
echo "# kernel sources $(python3 tools/source_fingerprint.py)" > $OUT
for seed in ${FUZZ_PARITY_SEEDS:-1001 2002 3003}; do
  echo "## tools/fuzz_parity.py 300 $seed" >> $OUT
  timeout 1500 python3 tools/fuzz_parity.py 300 $seed 2>&1 | grep -E "cases ok|mismatches|MISMATCH|Error|error" | head -20 >> $OUT
done
for seed in ${FUZZ_API_SEEDS:-31 32 33 34}; do
  echo "## tools/fuzz_api.py 500 $seed" >> $OUT
  timeout 1500 python3 tools/fuzz_api.py 500 $seed 2>&1 | tail -3 >> $OUT
done
echo "## tools/fuzz_last_bounce.py 120 ${FUZZ_LAST_SEED:-78}" >> $OUT
timeout 1500 python3 tools/fuzz_last_bounce.py 120 ${FUZZ_LAST_SEED:-78} 2>&1 | tail -3 >> $OUT
for seed in ${FUZZ_BVH_SEEDS:-7 99}; do
  echo "## tools/fuzz_bvh_build.py 8000 $seed" >> $OUT
  timeout 1500 python3 tools/fuzz_bvh_build.py 8000 $seed 2>&1 | tail -3 >> $OUT
done
cat $OUT
