#!/usr/bin/env python3
"""K sweep of the 256x256 GEMM kernel at fixed M, N (kernel-only time from the library's HIP
events): the slope is the cost of one K-tile, the intercept prologue + epilogue.  Dev tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi.api import Context

ctx = Context(0)
M, N = 4608, 27648
for K in (256, 512, 1024, 2048, 3072, 4096):
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    res = {}
    for name, fn in (("bf16", ctx.op_gemm_bf16), ("fp8", ctx.op_gemm_fp8)):
        for _ in range(3):
            fn(a, w)
        ctx.prof_begin(1)
        for _ in range(10):
            fn(a, w)
        n, ms, _ = ctx.prof_end()
        res[name] = ms / n
    fl = 2.0 * M * N * K
    print(f"K {K:5d}: bf16 {res['bf16']:.3f} ms ({fl / res['bf16'] / 1e9:7.1f} TF)  fp8 {res['fp8']:.3f} ms ({fl / res['fp8'] / 1e9:7.1f} TF)", flush=True)
