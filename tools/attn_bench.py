#!/usr/bin/env python3
"""Attention microbenchmark on the DiT's shapes: time per call and TFLOP/s (4 S^2 d), online-softmax op entry.  Dev tool.
(The blocks' pre-scaled form with and without the KV-split of the partial round: tools/attn_split_bench.py.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import flux2mi
from flux2mi.api import Context

SHAPES = [(4608, 24, "klein-4b 1024^2"), (768, 24, "klein-4b 256^2"), (4608, 32, "klein-9b 1024^2"), (4608, 48, "dev 1024^2"),
          (8704, 24, "klein-4b + 1024^2 reference"), (1536, 24, "klein-4b 512^2")]


def main():
    ctx = Context(0)
    for S, heads, name in SHAPES:
        d = heads * 128
        qkv = (torch.randn((S, 3 * d), device="cuda") * 0.7).to(torch.bfloat16)
        for _ in range(3):
            ctx.op_attention(qkv, heads)
        ctx.prof_begin(2)
        for _ in range(10):
            ctx.op_attention(qkv, heads)
        n, ms, fl = ctx.prof_end()
        print(f"{name:30s} S {S:5d} H {heads:2d}: {ms / n * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
