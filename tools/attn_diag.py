#!/usr/bin/env python3
"""Reads the s_memtime segment sums of the attention kernel's diagnostic build (libflux2mi_diag.so,
built with -DATTN_DIAG=1; never shipped): per-wave average cycles per KV tile spent in
{issue next-tile loads, QK^T, softmax, PV, stage + barrier}.  Shares, not lengths (stamps add fences)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FLUX2MI_LIB"] = "libflux2mi_diag.so"
import torch
import flux2mi
from flux2mi import _lib
from flux2mi.api import Context
ctx = Context(0)
h = C.CDLL(_lib.LIB_PATH)
h.flux2mi_debug_attn_stamps.argtypes = [C.POINTER(C.c_double), C.c_int]
for S, H in ((4608, 24), (8704, 48)):
    qkv = torch.randn((S, 3 * H * 128), device="cuda").to(torch.bfloat16)
    ctx.op_attention(qkv, H); ctx.sync()
    out = (C.c_double * 6)()
    h.flux2mi_debug_attn_stamps(out, 1)
    ctx.op_attention(qkv, H); ctx.sync()
    h.flux2mi_debug_attn_stamps(out, 1)
    tiles = out[5]   # sum over waves of tiles processed
    names = ["(unused)", "QK^T + prefetch issue", "softmax", "PV + LDS stores", "barrier"]
    tot = sum(out[:5])
    print(f"S{S} H{H}: per wave-tile cycles:", {n: round(out[i] / tiles) for i, n in enumerate(names)}, "total", round(tot / tiles))
