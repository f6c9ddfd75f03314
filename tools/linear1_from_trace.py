#!/usr/bin/env python3
"""The bench's roofline kernel seen by rocprofv3: from the kernel trace of `python bench.py --steps 1 ...` pick the launches of
single_blocks.linear1 — the gemm256s launch with the [q|k|mlp] grid (1728 tiles x 256 threads at Klein-4B 1024^2) and the V^T launch
that follows it — and average their durations the way bench.py averages its HIP-event brackets (`roofline.avg_launch_ms`: two
launches per block).  usage: linear1_from_trace.py <rocprof output dir> <out.json> [main-grid-size, default 442368]"""
import csv
import glob
import json
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
main_grid = int(sys.argv[3]) if len(sys.argv) > 3 else 442368
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
grid = lambda r: int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
gemms = [r for r in rows if "gemm256s_kernel" in r["Kernel_Name"]]
main, vt = [], []
for i, r in enumerate(gemms[:-1]):
    if grid(r) == main_grid:
        main.append(dur(r))
        vt.append(dur(gemms[i + 1]))
out = {"launch_pairs": len(main), "main_avg_ms": sum(main) / max(len(main), 1), "vt_avg_ms": sum(vt) / max(len(vt), 1),
       "avg_launch_ms": (sum(main) + sum(vt)) / max(2 * len(main), 1),
       "note": "average over the [q|k|mlp] launch and the V^T launch of every single block, as bench.py's roofline.avg_launch_ms"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
