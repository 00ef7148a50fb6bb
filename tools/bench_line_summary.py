#!/usr/bin/env python3
"""One line per workload of kept bench lines: value, ms per batch, ms per launch of every stage, shadow order, parity.  usage: bench_line_summary.py line.json ..."""
import json,sys
for f in sys.argv[1:]:
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,e); continue
    r=d['roofline']; print(f, d['value'], d['ms_per_step'], 'trav', round(r['stage_ms']['traverse']/r['stage_launches']['traverse'],3), d['config'].get('shadow_order',{}).get('order'), '|', d['config'].get('last_bounce_order',{}).get('mode'))
    for k,w in (d.get('workloads') or {}).items():
        r=w['roofline']; sl=r['stage_launches']
        print('   ',k,w['value'],w['ms_per_step'],{s:round(v/max(sl[s],1),3) for s,v in r['stage_ms'].items()}, w['config'].get('shadow_order',{}).get('order'), (w.get('parity_check') or {}).get('bitwise'))
