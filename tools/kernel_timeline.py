#!/usr/bin/env python3
"""Per-launch durations from a rocprofv3 --kernel-trace csv, from the LAST launch of a marker kernel on (e.g. one VAE decode:
marker latents_to_vae).  usage: kernel_timeline.py <rocprof output dir> <marker substring>   Dev tool."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"])
tot = 0.0
for r in rows[idx:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    print(f"{d:9.1f} us  grid {grid:>9s}  {r['Kernel_Name'][:90]}")
print(f"total {tot:.1f} us over {len(rows) - idx} launches")
