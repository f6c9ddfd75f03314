#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel name from *_counter_collection.csv files.
usage: pmc_summary.py <dir> [name-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if sub and sub not in name:
            continue
        key = (name[:90], row.get("Grid_Size", ""), row.get("Counter_Name", ""))
        acc[key][0] += float(row.get("Counter_Value", 0) or 0)
        acc[key][1] += 1
for (name, grid, ctr), (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{ctr:14s} avg {s / n:16.1f} n {n:5d} grid {grid:>9s} {name}")
