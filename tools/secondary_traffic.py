#!/usr/bin/env python3
"""HBM-side traffic per launch of the other hot kernels (the attention, LayerNorm + modulate) from the two rocprofv3 --pmc summaries of
tools/round_measure.sh (pmc_fetch_size.txt / pmc_write_size.txt, made by tools/pmc_summary.py over tools/one_forward.py: Klein-4B, 1024^2).
FETCH_SIZE doubled (gfx950: 128-B read requests counted as 64 B for 16-B/lane streams, MI355X_MICROARCH.md), WRITE_SIZE exact; KB -> bytes.
Writes the JSON bench.py reads for the `traffic` field of its roofline_secondary / roofline_hbm entries.
usage: secondary_traffic.py <pmc_fetch_size.txt> <pmc_write_size.txt> <out.json>"""
import json
import re
import sys

KERNELS = {  # bench prof class -> (kernel-name substring, grid in threads at Klein-4B 1024^2, algorithmic bytes per launch)
    "2": ("attn64b_kernel<false, false>", 110592, 4608 * 3072 * 2 * 4),            # q, k, v read + o written, bf16
    "5": ("layernorm_modulate_kernel", 1179648, 4608 * 3072 * (4 + 2)),             # f32 row in, bf16 row out
}


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.match(rf"{counter}\s+avg\s+([\d.]+)\s+n\s+(\d+)\s+grid\s+(\d+)\s+(.*)$", line.strip())
        if m:
            out[(m.group(4), int(m.group(3)))] = (float(m.group(1)), int(m.group(2)))
    return out


fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/one_forward.py klein-4b 2, per kernel and grid "
                 "(tools/pmc_summary.py); FETCH_SIZE doubled (gfx950), WRITE_SIZE exact", "kernels": {}}
for cls, (name, grid, alg) in KERNELS.items():
    f = [v for (k, g), v in fetch.items() if name in k and g == grid]
    w = [v for (k, g), v in write.items() if name in k and g == grid]
    assert f and w, name
    res["kernels"][cls] = {"kernel": name, "fetch_size_kb": f[0][0], "write_size_kb": w[0][0], "launches": f[0][1],
                           "traffic_bytes_per_launch": (2 * f[0][0] + w[0][0]) * 1024, "algorithmic_bytes_per_launch": alg}
print(json.dumps(res, indent=1))
json.dump(res, open(sys.argv[3], "w"), indent=1)
