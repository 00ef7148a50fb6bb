#!/bin/bash
# quick multi-workload bench summary (developer tool): bash tools/bench_all.sh [extra bench.py args]
for w in darkcornell darkcornell_mis veachmis pbrtest; do
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-workloads --workload $w "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:44], '| Mrays/s', d['value'], '| Msamples/s', round(d['samples_per_s']/1e6,1), '| ms/step', d['ms_per_step'], d['roofline']['stage_ms'])"
done
