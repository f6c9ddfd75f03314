#!/usr/bin/env python3
"""A few launches of the attention kernel at the production shape (S = 4608, 24 heads) for
`rocprofv3 --pmc` passes.  Dev tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi.api import Context
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
H = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ctx = Context(0)
qkv = torch.randn((S, 3 * H * 128), device="cuda").to(torch.bfloat16)
for _ in range(4):
    out = ctx.op_attention(qkv, H)
ctx.sync()
print("ok", float(out.float().abs().mean()))
