#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace directory, WARM launches only: the first SKIP launches
of every kernel are dropped (cold caches, clocks still ramping, first touches), so that the average can be compared with
a bench step.  Usage: tools/warm_kernel_stats.py TRACE_DIR [SKIP=2] > kernel_stats.csv"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = defaultdict(list)
for r in rows:
    per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["Name", "Launches", "SkippedColdLaunches", "WarmCalls", "AverageNs", "MinNs", "MaxNs", "TotalWarmNs"])
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    warm = v[skip:] if len(v) > skip else []
    if warm:
        w.writerow([k, len(v), skip, len(warm), sum(warm) / len(warm), min(warm), max(warm), sum(warm)])
    else:
        w.writerow([k, len(v), len(v), 0, "", "", "", 0])
