python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "production or knobs or image_parity or baseline_configs" 2>&1 | tail -15
for wl in veachmis pbrtest furnace; do
  for tt in 0 1; do
    RPT_TOP_TREE=$tt python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-readback --no-extra-workloads 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl top_tree=$tt', d['value'], d['ms_per_step'], d['roofline']['stage_ms'], d['parity_check']['bitwise'])"
  done
done
