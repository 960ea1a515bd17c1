#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs (one directory per pass) per kernel: tools/pmc_summary.py DIR [WORKLOAD_KEY] > summary.json
"_meta" records which sources the counters were taken from (tools/profile_hash.py) and the workload's key
(g<grid>_<nx>x<ny>x<spp>_d<depth>), which bench.py compares with its own."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import profile_hash  # noqa: E402

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k][row["Counter_Name"]] += 1
out = {"_meta": {"source_hash": profile_hash.kernel_source_hash(), "workload": sys.argv[2] if len(sys.argv) > 2 else None}}
for k in acc:
    out[k] = {c: {"sum": v, "dispatches": calls[k][c], "per_dispatch": v / calls[k][c]} for c, v in sorted(acc[k].items())}
json.dump(out, sys.stdout, indent=1)
