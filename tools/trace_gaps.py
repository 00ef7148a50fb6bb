"""Busy time and gaps of the LAST batch (from its k_generate_first on) in a rocprofv3 kernel_trace.csv.
usage: trace_gaps.py KERNEL_TRACE_CSV"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_generate_first")]
for which, label in ((-2, "second to last batch"), (-1, "last batch")):
    if len(starts) < -which: continue
    lo = starts[which]; hi = starts[which + 1] if which != -1 else len(rows)
    batch = [r for r in rows[lo:hi] if not r[2].startswith("__amd")]
    busy = sum(e - s for s, e, _ in batch)
    span = batch[-1][1] - batch[0][0]
    gaps = [batch[i + 1][0] - batch[i][1] for i in range(len(batch) - 1)]
    print(f"{label}: {len(batch)} kernels, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, gaps {sum(gaps) / 1e3:.1f} us "
          f"({100.0 * sum(gaps) / span:.1f} %), largest gap {max(gaps) / 1e3:.1f} us, median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us")
    if which == -2 and hi < len(rows): print(f"   gap to the next batch: {(rows[hi][0] - batch[-1][1]) / 1e3:.1f} us")
