#!/usr/bin/env python3
"""Pre-scaled attention (the model's form) on the DiT's shapes: the fixed zero offset (score bound known and <= 80) against the
per-query data offset (no bound) of attention64b.hip, on unit-RMS q / k rows as the QK RMSNorm leaves them.  Three interleaved
rounds; time per call from the library's HIP events.  Dev tool."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import flux2mi
from flux2mi.api import Context

SHAPES = [(4608, 24, "klein-4b 1024^2"), (4608, 32, "klein-9b 1024^2"), (4608, 48, "dev 1024^2"), (8704, 24, "klein-4b + 1024^2 reference"),
          (768, 24, "klein-4b 256^2")]


def main():
    ctx = Context(0)
    c = math.log2(math.e) / math.sqrt(128.0)
    for S, heads, name in SHAPES:
        d = heads * 128
        x = torch.randn((S, 3 * d), device="cuda")
        q = x[:, :d].view(S, heads, 128)
        k = x[:, d:2 * d].view(S, heads, 128)
        q = q / q.pow(2).mean(-1, keepdim=True).sqrt() * c
        k = k / k.pow(2).mean(-1, keepdim=True).sqrt()
        qkv = torch.cat([q.reshape(S, d), k.reshape(S, d), x[:, 2 * d:]], dim=1).to(torch.bfloat16)
        bound = 128.0 * c
        res = {"fixed": [], "data": []}
        for _ in range(3):
            for label, b in (("fixed", bound), ("data", 0.0)):
                for _ in range(3):
                    ctx.op_attention_prescaled(qkv, heads, b)
                ctx.prof_begin(2)
                for _ in range(10):
                    ctx.op_attention_prescaled(qkv, heads, b)
                n, ms, fl = ctx.prof_end()
                res[label].append(ms / n * 1e3)
        fl = 4.0 * S * S * 128 * heads
        f, dd = min(res["fixed"]), min(res["data"])
        print(f"{name:30s} S {S:5d} H {heads:2d}: fixed offset {f:7.1f} us ({fl / f / 1e6:6.1f} TF/s)   data offset {dd:7.1f} us ({fl / dd / 1e6:6.1f} TF/s)   "
              f"+{(dd / f - 1) * 100:.1f} %", flush=True)


if __name__ == "__main__":
    main()
