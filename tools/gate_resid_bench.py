#!/usr/bin/env python3
"""Kernel time of the gate+residual GEMM (linear2 / mlp_out shapes), library HIP events.  Dev tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi.api import Context
ctx = Context(0)
for (M, N, K) in [(4608, 3072, 12288), (4608, 3072, 9216)]:
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * 0.02).to(torch.bfloat16)
    gate = torch.randn((N,), device="cuda")
    x = torch.randn((M, N), device="cuda")
    for _ in range(3):
        ctx.op_gemm_gate_resid(a, w, gate, x)
    ctx.prof_begin(1)
    for _ in range(20):
        ctx.op_gemm_gate_resid(a, w, gate, x)
    n, ms, _ = ctx.prof_end()
    print(f"K {K}: {ms / n * 1e3:.1f} us  {2.0 * M * N * K / (ms / n) / 1e9:.0f} TF", flush=True)
