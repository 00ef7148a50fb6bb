"""The kernels of the LAST batch of a bench.py run in launch order with their durations, from rocprofv3's kernel_trace.csv.
usage: trace_sequence.py KERNEL_TRACE_CSV"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_generate_first")]
for s, e, name in rows[starts[-1]:]:
    print(f"{(e - s) / 1e3:10.1f} us  {name.split('(')[0][:90]}")
