python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "production or knobs" 2>&1 | tail -4
V=rust-path-tracer_amd/lib/variants
run() { # label, env...
  local label=$1; shift
  for wl in veachmis pbrtest; do
    env "$@" timeout 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-readback --no-extra-workloads --no-parity-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', '$label', d['value'], d['ms_per_step'], {k: v for k, v in d['roofline']['stage_ms'].items() if k in ('traverse', 'shadow')})"
  done
}
run base RPT_TOP_TREE=0
run top RPT_TOP_TREE=1
run top_k512 RPT_TOP_TREE=1 RPT_TOP_PAIRS=512
run top_bfs RPT_TOP_TREE=1 RPT_TOP_ORDER=1
run top_pct150 RPT_TOP_TREE=1 RPT_HIP_LIB=$V/pct150.so
run top_pct250 RPT_TOP_TREE=1 RPT_HIP_LIB=$V/pct250.so
run top_pct60 RPT_TOP_TREE=1 RPT_HIP_LIB=$V/pct60.so
run top_blocks512 RPT_TOP_TREE=1 RPT_TOP_BLOCKS=512
