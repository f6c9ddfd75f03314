#!/usr/bin/env python3
"""Matrix-pipe and LDS counters of the hot kernels from two rocprofv3 --pmc passes over tools/one_forward.py (see tools/round_measure.sh):
per kernel and grid, SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES x 4 SIMDs) = the share of SIMD-cycles with the matrix pipe busy
while the CU has a wave, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = the share of LDS cycles lost to bank conflicts, and the wait
buckets as shares of SQ_WAVE_CYCLES.  usage: pmc_sq.py <dir of pass 1> <dir of pass 2>"""
import collections
import csv
import glob
import sys


def load(d):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            key = (row.get("Kernel_Name", "")[:70], row.get("Grid_Size", ""), row.get("Counter_Name", ""))
            acc[key][0] += float(row.get("Counter_Value", 0) or 0)
            acc[key][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}, {k[:2]: v[1] for k, v in acc.items()}


def main():
    vals, counts = {}, {}
    for d in sys.argv[1:3]:
        v, c = load(d)
        vals.update(v)
        counts.update(c)
    kernels = sorted({k[:2] for k in vals}, key=lambda kg: -vals.get(kg + ("SQ_BUSY_CU_CYCLES",), 0.0) * counts.get(kg, 0))
    for kg in kernels:
        g = lambda c: vals.get(kg + (c,), float("nan"))
        if not any(t in kg[0] for t in ("gemm", "attn", "conv2d", "gn_", "softmax", "groupnorm")):
            continue
        print(f"{kg[0]}  grid {kg[1]}  launches {counts[kg]}")
        print(f"    MFMA busy / (CU busy x 4 SIMDs) {g('SQ_VALU_MFMA_BUSY_CYCLES') / (4.0 * g('SQ_BUSY_CU_CYCLES')):.3f}   "
              f"LDS bank-conflict share {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1.0):.4f}   "
              f"wait_any / wave_cycles {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}   wait_inst_any / wave_cycles {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}")
        print("    " + "  ".join(f"{c} {g(c):.4g}" for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                                            "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")))


if __name__ == "__main__":
    main()
