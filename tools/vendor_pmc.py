#!/usr/bin/env python3
"""A few launches of the 256x256 GEMM here and of hipBLASLt (torch.matmul) on the same operands, for rocprofv3 --pmc passes
(counters per kernel side by side).  Dev tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi.api import Context

ctx = Context(0)
for M, N, K in ((8192, 8192, 8192), (4608, 27648, 3072), (4608, 3072, 12288)):
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    c = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    c2 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    for _ in range(6):
        with torch.cuda.stream(ctx.stream):
            ctx.op_gemm_bf16_out_bf16(a, w, c)
        ctx.sync()
    for _ in range(6):
        torch.matmul(a, w.t(), out=c2)
    torch.cuda.synchronize()
    print(M, N, K, float((c.float() - c2.float()).abs().max()), flush=True)
