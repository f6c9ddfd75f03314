#!/usr/bin/env python3
"""Does a partial last round of workgroups cost a whole round?  M 4608 (18 row tiles) x N = 256 n, K 3072: time per tile for
tile counts just under / over multiples of the 256 CUs.  Dev tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi.api import Context

ctx = Context(0)
M, K = 4608, 3072
a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
for n in (71, 72, 85, 86, 90, 96, 99, 100, 113, 114):
    N = 256 * n
    w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    ts = []
    for r in range(5):
        with torch.cuda.stream(ctx.stream):
            for _ in range(3):
                ctx.op_gemm_bf16(a, w)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ctx.op_gemm_bf16(a, w)
            e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 10)
    med = sorted(ts)[2]
    tiles = 18 * n
    print(f"n {n:3d}: {tiles:5d} tiles = {tiles / 256:5.2f} rounds  {med * 1e3:7.1f} us  {med * 1e3 / tiles * 256:6.1f} us per 256 tiles  {2.0 * M * N * K / med / 1e9:6.0f} TF", flush=True)
