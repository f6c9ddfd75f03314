#!/usr/bin/env python3
"""The attention's KV-split of the partial last round by head count (attention64b.hip, AttnSplit): the blocks' pre-scaled form
(flux2mi_op_attention_prescaled: bf16 Q K^T) and the fp8 models' (flux2mi_op_attention_qk8: e4m3 Q K^T) at S 4608, unit-RMS q / k; HIP-event
time of the attention launch(es) alone (the op's V^T transpose is outside the bracket; the combine kernel is inside).  Run it once with the
product library and once with FLUX2MI_LIB=libflux2mi_dev.so FLUX2MI_ATTN_SPLIT=0 on the same box: profiles/r04_k_attention_split_by_heads.txt.
Dev tool (GPU)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import flux2mi  # noqa: F401
from flux2mi.api import Context

ctx = Context(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
c = math.log2(math.e) / math.sqrt(128.0)
print(f"# S {S}; library {os.environ.get('FLUX2MI_LIB', 'libflux2mi.so')}, FLUX2MI_ATTN_SPLIT={os.environ.get('FLUX2MI_ATTN_SPLIT', '(default on)')}")
print("# heads items rounds pieces | bf16 QK^T us | e4m3 QK^T us")
for heads in (8, 16, 24, 32, 40, 48, 56, 64):
    d = heads * 128
    g = torch.Generator(device="cuda").manual_seed(heads)
    x = torch.randn((S, 3 * d), generator=g, device="cuda")
    q = x[:, :d].view(S, heads, 128); k = x[:, d:2 * d].view(S, heads, 128)
    q = q / q.pow(2).mean(-1, keepdim=True).sqrt(); k = k / k.pow(2).mean(-1, keepdim=True).sqrt()
    qkv = torch.cat([(q * c).reshape(S, d), k.reshape(S, d), x[:, 2 * d:]], 1).to(torch.bfloat16)
    q8 = ((q * c).reshape(S, d) * 2.0 ** 7).clamp(-448, 448).to(torch.float8_e4m3fn)
    k8 = (k.reshape(S, d) * 2.0 ** 4).clamp(-448, 448).to(torch.float8_e4m3fn)
    v = x[:, 2 * d:].to(torch.bfloat16).contiguous()
    bound = 1.5 * 128.0 * c
    f, n_cu = ctx.op_attention_split_info(S, heads)
    items = ((S + 255) // 256) * heads
    out = []
    for fn in (lambda: ctx.op_attention_prescaled(qkv, heads, bound), lambda: ctx.op_attention_qk8(q8, k8, v, heads, 7, 4, bound)):
        for _ in range(3):
            fn()
        ctx.prof_begin(2)
        for _ in range(10):
            fn()
        n, ms, _ = ctx.prof_end()
        out.append(ms / n * 1e3)
    print(f"{heads:3d} {items:5d} {items / n_cu:6.2f} {f:2d} | {out[0]:8.1f} | {out[1]:8.1f}", flush=True)
