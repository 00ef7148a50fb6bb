for v in "$@"; do
  echo "== $v"
  RPT_HIP_LIB=$PWD/rust-path-tracer_amd/lib/librpt_hip_$v.so RPT_STAGE_TIMING=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"roofline\"][\"stage_ms\"])"
done
