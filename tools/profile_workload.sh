#!/bin/bash
# Judged evidence of ONE bench workload on the GPU box, into gpurun_out/profile/:
#   ${TAG}_${WL}_bench.json          the bench line (python3 bench.py --workload WL ...)
#   ${TAG}_${WL}_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command (--no-cpu-baseline)
#   ${TAG}_${WL}_kernel_stats_timed.txt  the same trace restricted to the timed batches (tools/trace_region.py) and
#   ${TAG}_${WL}_bench_under_rocprof.json the bench line of that profiled run: its roofline.avg_launch_ms is the figure to compare
#   ${TAG}_${WL}_pmc_{FETCH_SIZE,WRITE_SIZE}.txt   per-kernel per-launch averages, one --pmc pass each (--kernel-trace only)
#   ${TAG}_${WL}_pmc_sq.txt, _pmc_cache.txt        SQ issue / lane utilisation, TCP / TCC hit passes;  _pmc_sq3 / _pmc_sq4: VALU instruction classes (f32 / f64 + int64)
#   ${TAG}_${WL}_pmc_ta.txt                        texture-address unit busy cycles (what bounds the global-memory walks)
#   traffic_${WL}.json               HBM bytes per launch per stage, VALU issue figures of the SQ pass + source fingerprint (bench.py reads profiles/traffic_*.json)
# usage: tools/profile_workload.sh TAG WORKLOAD [bench args, e.g. --steps 4]     env: SKIP_PMC=1 keeps only bench + stats
set -u
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --workload $WL --no-extra-workloads"
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $B --no-cpu-baseline "$@" > "$OUT/${TAG}_${WL}_stats_run.log" 2>&1
cp "$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_${WL}_kernel_stats.csv"
python3 "$ROOT/tools/trace_region.py" "$(find /tmp/prof_stats -name '*kernel_trace.csv' | head -1)" "$@" > "$OUT/${TAG}_${WL}_kernel_stats_timed.txt" 2>&1
grep '"metric"' "$OUT/${TAG}_${WL}_stats_run.log" | tail -1 > "$OUT/${TAG}_${WL}_bench_under_rocprof.json"
if [ "${SKIP_PMC:-0}" != "1" ]; then
  pass() {   # name, counters...
    local name=$1; shift
    rm -rf /tmp/prof_p
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/prof_p -- python3 $B --steps 2 --warmup 0 --no-cpu-baseline > "$OUT/${TAG}_${WL}_pmc_${name}.log" 2>&1
    local f; f=$(find /tmp/prof_p -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then
      cp "$f" "/tmp/${name}.csv"
      { echo "# rocprofv3 --pmc $* --kernel-trace -- python3 bench.py --workload $WL --steps 2 --warmup 0 --no-cpu-baseline ; per-launch averages"; python3 "$ROOT/tools/pmc_summary.py" "$f"; } > "$OUT/${TAG}_${WL}_pmc_${name}.txt"
      rm -f "$OUT/${TAG}_${WL}_pmc_${name}.log"
    fi
  }
  pass FETCH_SIZE FETCH_SIZE
  pass WRITE_SIZE WRITE_SIZE
  python3 "$ROOT/tools/traffic_from_pmc.py" /tmp/FETCH_SIZE.csv /tmp/WRITE_SIZE.csv "$WL" "$OUT/traffic_${WL}.json" \
    "profiles/${TAG}_${WL}_pmc_FETCH_SIZE.txt, profiles/${TAG}_${WL}_pmc_WRITE_SIZE.txt (separate --pmc passes, bench.py --workload $WL --steps 2 --warmup 0 --no-cpu-baseline)" > /dev/null
  pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
  pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
  pass sq3 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
  pass sq4 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64
  pass cache TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
  pass ta TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum
  [ -s "$OUT/${TAG}_${WL}_pmc_sq.txt" ] && python3 "$ROOT/tools/valu_from_pmc.py" "$OUT/${TAG}_${WL}_pmc_sq.txt" "$OUT/traffic_${WL}.json" > /dev/null
  [ -s "$OUT/${TAG}_${WL}_pmc_sq3.txt" ] && [ -s "$OUT/${TAG}_${WL}_pmc_sq4.txt" ] && python3 "$ROOT/tools/valu_mix_from_pmc.py" "$OUT/${TAG}_${WL}_pmc_sq3.txt" "$OUT/${TAG}_${WL}_pmc_sq4.txt" "$OUT/traffic_${WL}.json" > /dev/null
  [ -s "$OUT/${TAG}_${WL}_pmc_ta.txt" ] && python3 "$ROOT/tools/ta_from_pmc.py" "$OUT/${TAG}_${WL}_pmc_ta.txt" "$OUT/traffic_${WL}.json" > /dev/null
fi
# the bench line last: it reports `traffic` only from a traffic_*.json measured on exactly these kernel sources
[ -s "$OUT/traffic_${WL}.json" ] && cp "$OUT/traffic_${WL}.json" "$ROOT/profiles/traffic_${WL}.json"
python3 $B "$@" 2>"$OUT/${TAG}_${WL}_bench.err" | tail -1 > "$OUT/${TAG}_${WL}_bench.json"
head -8 "$OUT/${TAG}_${WL}_kernel_stats.csv" | cut -c1-200
cat "$OUT/${TAG}_${WL}_bench.json" | cut -c1-600
