#!/usr/bin/env python3
"""Does the OPERAND LAYOUT pace the 256x256 GEMM's K loop?  Times the same products with
  rm      plain row-major operands (row stride = K elements)
  pad     row-major with every row padded by 128 B (lda = ldw = K + 64)
  wblk    W in the K-blocked layout [K/64][N][64] (a workgroup's 256-row slice per K-tile = one contiguous 32 KB block)
  ablk    A K-blocked, W row-major
  blk     both K-blocked
interleaved in one process (rounds x variants), values checked against torch on the first round.  Dev tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import flux2mi
from flux2mi.api import Context

SHAPES = [(4608, 27648, 3072, "linear1"), (4608, 3072, 12288, "linear2"), (4608, 3072, 9216, "mlp_out"),
          (4608, 9216, 3072, "qkv"), (8192, 8192, 8192, "square8k")]


def kblock(x):
    r, k = x.shape
    return x.view(r, k // 64, 64).permute(1, 0, 2).contiguous()


def pad(x):
    r, k = x.shape
    y = torch.zeros((r, k + 64), dtype=x.dtype, device=x.device)
    y[:, :k] = x
    return y


def main():
    ctx = Context(0)
    rounds = 5
    for M, N, K, name in SHAPES:
        a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
        w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
        ab, wb, ap, wp = kblock(a), kblock(w), pad(a), pad(w)
        variants = {
            "rm": lambda: ctx.op_gemm_bf16_ex(a, K, 0, w, K, 0, M, N, K),
            "pad": lambda: ctx.op_gemm_bf16_ex(ap, K + 64, 0, wp, K + 64, 0, M, N, K),
            "wblk": lambda: ctx.op_gemm_bf16_ex(a, K, 0, wb, 64, N * 128, M, N, K),
            "ablk": lambda: ctx.op_gemm_bf16_ex(ab, 64, M * 128, w, K, 0, M, N, K),
            "blk": lambda: ctx.op_gemm_bf16_ex(ab, 64, M * 128, wb, 64, N * 128, M, N, K),
        }
        ref = torch.matmul(a.float(), w.float().t())
        times = {k: [] for k in variants}
        for k, fn in variants.items():
            out = fn()
            err = float((out - ref).norm() / ref.norm())
            assert err < 5e-3, (name, k, err)
        del ref
        for r in range(rounds):
            for k, fn in variants.items():
                with torch.cuda.stream(ctx.stream):
                    for _ in range(2):
                        fn()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(8):
                        fn()
                    e.record()
                torch.cuda.synchronize()
                times[k].append(s.elapsed_time(e) / 8)
        fl = 2.0 * M * N * K
        line = f"{name:9s} M{M} N{N} K{K}: "
        for k in variants:
            med = sorted(times[k])[len(times[k]) // 2]
            line += f"{k} {med:.3f} ms {fl / med / 1e9:6.0f} TF | "
        print(line, flush=True)


if __name__ == "__main__":
    main()
