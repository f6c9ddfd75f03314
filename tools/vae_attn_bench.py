#!/usr/bin/env python3
"""The VAE mid block's streaming attention alone (flux2mi_op_vae_attention = csrc/vae_attention.hip behind a V transpose), HIP events
around the kernel (class 8): P = 16384 (1024^2 decode), 24576 (1536x1024), 65536 (2048^2); head dim 512 / 384.  Dev tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi.api import Context

ctx = Context(0)
for C in (512, 384):
    for P in (4096, 16384, 24576, 65536):
        g = torch.Generator(device="cuda").manual_seed(P + C)
        q, k, v = (torch.randn((P, C), generator=g, device="cuda").to(torch.bfloat16) for _ in range(3))
        for _ in range(2):
            ctx.op_vae_attention(q, k, v)
        ctx.prof_begin(8)
        for _ in range(5):
            ctx.op_vae_attention(q, k, v)
        n, ms, fl = ctx.prof_end()
        print(f"vae attention C {C} P {P}: {ms / n * 1e3:.1f} us per launch, {fl / ms / 1e9:.0f} TFLOP/s ({fl / ms / 1e9 / 2500:.3f} of 2.5 PF), "
              f"{(P + 63) // 64} workgroups")
