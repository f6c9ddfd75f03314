#!/usr/bin/env python3
"""The blocks' attention form (pre-scaled q, fixed zero offset) against a torch fp32 reference and against the 8-wave kernel
(dev library, FLUX2MI_ATTN64=0 in a child process), then timing.  Dev tool."""
import math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi.api import Context

SHAPES = [(64, 1), (100, 2), (128, 2), (192, 1), (200, 3), (256, 3), (330, 1), (704, 2), (768, 24), (1153, 2), (4112, 24), (4608, 24), (4608, 32), (8704, 24)]


def make(S, heads, seed):
    g = torch.Generator().manual_seed(seed)
    d = heads * 128
    qkv = torch.randn((S, 3 * d), generator=g)
    # unit-RMS q and k rows per head (what the QK RMSNorm leaves), q pre-scaled
    q = qkv[:, :d].view(S, heads, 128)
    k = qkv[:, d:2 * d].view(S, heads, 128)
    q = q / q.pow(2).mean(-1, keepdim=True).sqrt()
    k = k / k.pow(2).mean(-1, keepdim=True).sqrt()
    c = math.log2(math.e) / math.sqrt(128.0)
    qkv = torch.cat([(q * c).reshape(S, d), k.reshape(S, d), qkv[:, 2 * d:]], dim=1).to(torch.bfloat16)
    return qkv, c


def ref(qkv, heads, c):
    S = qkv.shape[0]
    d = heads * 128
    q = qkv[:, :d].float().view(S, heads, 128).permute(1, 0, 2) / c
    k = qkv[:, d:2 * d].float().view(S, heads, 128).permute(1, 0, 2)
    v = qkv[:, 2 * d:].float().view(S, heads, 128).permute(1, 0, 2)
    o = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(128.0), dim=-1) @ v
    return o.permute(1, 0, 2).reshape(S, d)


def main():
    ctx = Context(0)
    bound = 128.0 * math.log2(math.e) / math.sqrt(128.0)   # |q.k| <= 128 for unit-RMS rows
    for S, heads in SHAPES:
        qkv, c = make(S, heads, S + heads)
        x = qkv.cuda()
        out = ctx.op_attention_prescaled(x, heads, bound).float()
        if S <= 4608 and heads <= 24:
            r = ref(x, heads, c)
            err = float((out - r).norm() / r.norm())
            mx = float((out - r).abs().max())
        else:
            err = mx = float("nan")
        same = torch.equal(out, ctx.op_attention_prescaled(x, heads, bound).float())
        print(f"S {S:5d} H {heads:2d}: rel {err:.2e} max {mx:.2e} finite {bool(torch.isfinite(out).all())} rerun-identical {same}", flush=True)
        if S >= 768:
            for _ in range(3):
                ctx.op_attention_prescaled(x, heads, bound)
            ctx.prof_begin(2)
            for _ in range(10):
                ctx.op_attention_prescaled(x, heads, bound)
            n, ms, fl = ctx.prof_end()
            print(f"    {ms / n * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
