#!/usr/bin/env python3
"""Per-kernel statistics over WARM dispatches only, from a rocprofv3 --kernel-trace run.

usage: tools/warm_kernel_stats.py <rocprofv3 output dir> [--drop 5] [--min-calls 1] > profiles/rNN_<tag>_kernel_stats.csv

rocprofv3's own `--stats` table averages every dispatch of a kernel, the first ones included: a cold code object, first-touch
page faults of the scratch and a clock that is still ramping (VERDICT r03 "What's weak" 5: 6.75 ms average against 6.32 ms
in the bench line, only the minimum agreed).  This script reads the per-dispatch `*_kernel_trace.csv` of the same run, drops
the first `--drop` dispatches of each kernel (in start order) and prints the table in `--stats`' own column layout, with the
number of dropped dispatches and the all-dispatch average next to it."""
import argparse, collections, csv, glob, math, os, sys

ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("--drop", type=int, default=5)
ap.add_argument("--min-calls", type=int, default=1)
a = ap.parse_args()
files = sorted(glob.glob(os.path.join(a.root, "**", "*_kernel_trace.csv"), recursive=True))
if not files:
    sys.exit(f"no *_kernel_trace.csv under {a.root}")
rows = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
out = []
for name, ev in rows.items():
    ev.sort()
    dur_all = [e - s for s, e in ev]
    warm = dur_all[a.drop:] if len(dur_all) > a.drop else dur_all
    if len(warm) < a.min_calls:
        continue
    mean = sum(warm) / len(warm)
    sd = math.sqrt(sum((d - mean) ** 2 for d in warm) / len(warm))
    out.append((sum(warm), name, len(warm), mean, min(warm), max(warm), sd, len(dur_all) - len(warm), sum(dur_all) / len(dur_all)))
total = sum(o[0] for o in out) or 1
w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
print(f"# warm dispatches only: the first {a.drop} dispatches of every kernel dropped (tools/warm_kernel_stats.py on the *_kernel_trace.csv of "
      f"`rocprofv3 --kernel-trace --stats`); AllCallsAverageNs = rocprofv3's own --stats average over every dispatch")
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "DroppedCalls", "AllCallsAverageNs"])
for tot, name, n, mean, lo, hi, sd, dropped, mean_all in sorted(out, reverse=True):
    w.writerow([name, n, tot, round(mean, 3), round(100.0 * tot / total, 2), lo, hi, round(sd, 3), dropped, round(mean_all, 3)])
