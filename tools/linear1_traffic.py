#!/usr/bin/env python3
"""HBM-side traffic of single_blocks.linear1 from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
tools/one_forward.py.  linear1 is two launches since round 1 session 2: [q|k|mlp] (1728 workgroups) and V^T
(the 224-row-tile launch right after it).  Prints / writes the JSON bench.py reports as roofline.traffic:
bytes per launch = (sum over both launches) / 2, FETCH_SIZE doubled (gfx950 note in MI355X_MICROARCH.md).
usage: linear1_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, sys

def per_dispatch(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])))
    rows.sort()
    return rows

def linear1_pairs(rows):
    """value sums of (main, vt) pairs: a 256x256 GEMM dispatch of 1728 workgroups followed by the next GEMM dispatch
    (gemm256s_kernel: 256 threads per workgroup; the round-1 / early round-2 gemm_bf16_256x256_kernel: 512)"""
    out = []
    gemm = [r for r in rows if "gemm256s_kernel" in r[1] or "gemm_bf16_256x256_kernel" in r[1]]
    for a, b in zip(gemm, gemm[1:]):
        if a[2] in (1728 * 256, 1728 * 512):
            out.append((a[3], b[3], b[2] // (a[2] // 1728)))
    return out

fetch = linear1_pairs(per_dispatch(sys.argv[1], "FETCH_SIZE"))
write = linear1_pairs(per_dispatch(sys.argv[2], "WRITE_SIZE"))
assert fetch and write, "no linear1 launches found"
f_main = sum(x[0] for x in fetch) / len(fetch); f_vt = sum(x[1] for x in fetch) / len(fetch)
w_main = sum(x[0] for x in write) / len(write); w_vt = sum(x[1] for x in write) / len(write)
S, d = 4608, 3072
alg = (S * d * 2 + 8 * d * d * 2 + S * 2 * d * 2 + S * 3 * d * 2) + (S * d * 2 + d * d * 2 + S * d * 2)   # A + W + outputs, both launches
res = {
    "kernel": "gemm256s_kernel, single_blocks.linear1 = [q|k|mlp] launch (1728 workgroups) + V^T launch (%d workgroups)" % fetch[0][2],
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/one_forward.py klein-4b 2; average of %d linear1 pairs" % len(fetch),
    "fetch_size_kb": {"main": f_main, "vt": f_vt}, "write_size_kb": {"main": w_main, "vt": w_vt},
    "correction": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B for 16-B/lane streams (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE exact",
    "traffic_bytes_per_launch": ((f_main + f_vt) * 2 + (w_main + w_vt)) * 1024 / 2,
    "algorithmic_bytes_per_launch": alg / 2,
}
print(json.dumps(res, indent=1))
json.dump(res, open(sys.argv[3], "w"), indent=1)
