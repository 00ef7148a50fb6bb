"""Per-kernel average duration over the TIMED batches of a bench.py run, from rocprofv3's kernel_trace.csv.
`--kernel-trace --stats` averages over every launch of the process — set-up and warm-up batches at a cold clock, the
read-back loop — while bench.py's `roofline.avg_launch_ms` covers the timed region only; this is the like-for-like figure.
A batch starts at a k_generate_first; bench.py runs 2 set-up + W warm-up batches, then the K timed ones.
usage: trace_region.py KERNEL_TRACE_CSV --warmup W --steps K"""
import argparse, csv, collections

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--setup", type=int, default=2)
a, _ = ap.parse_known_args()
rows = []
with open(a.trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
batch, first, last = -1, a.setup + a.warmup, a.setup + a.warmup + a.steps
acc = collections.OrderedDict()
for s, e, name in rows:
    if name.startswith("k_generate_first"):
        batch += 1
    if first <= batch < last:
        short = name.split("(")[0]
        n, t = acc.get(short, (0, 0))
        acc[short] = (n + 1, t + (e - s))
print(f"# batches {first}..{last - 1} of the run (after {a.setup} set-up + {a.warmup} warm-up): the {a.steps} timed ones")
print(f"# {'kernel':70s} {'launches':>8s} {'avg us':>10s} {'total ms':>10s}")
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:72s} {n:8d} {t / n / 1e3:10.1f} {t / 1e6:10.3f}")
