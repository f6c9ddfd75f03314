#!/usr/bin/env python3
"""Microbenchmark of the attention kernel at the DiT's shapes (random data).  Dev tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi.api import Context

def time_fn(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

ctx = Context(0)
for S, H in [(4608, 24), (4608, 32), (4608, 48), (8704, 48), (768, 24)]:
    d = H * 128
    qkv = torch.randn((S, 3 * d), device="cuda").to(torch.bfloat16)
    with torch.cuda.stream(ctx.stream):
        ms = time_fn(lambda: ctx.op_attention(qkv, H))
    q, k, v = [x.reshape(S, H, 128).permute(1, 0, 2).unsqueeze(0).contiguous() for x in qkv.split(d, dim=1)]
    ms_t = time_fn(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
    fl = 4.0 * S * S * d
    print(f"S{S} H{H}: ours {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF (incl. V^T transpose) | torch sdpa {ms_t:7.3f} ms {fl / ms_t / 1e9:7.1f} TF", flush=True)
