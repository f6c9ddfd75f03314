#!/usr/bin/env python3
"""Microbenchmark of the GEMM kernels on the DiT's production shapes (random data), with
hipBLASLt (torch.matmul) timed beside them as the vendor reference point.  Dev tool."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import flux2mi
from flux2mi.api import Context

SHAPES = [  # (M, N, K, name)
    (4608, 27648, 3072, "single.linear1"),
    (4608, 3072, 12288, "single.linear2"),
    (4608, 9216, 3072, "double.qkv"),
    (4608, 3072, 3072, "double.proj"),
    (4608, 18432, 3072, "double.mlp_in"),
    (4608, 3072, 9216, "double.mlp_out"),
    (4096, 4096, 4096, "square4k"),
    (8192, 8192, 8192, "square8k"),
]


SHAPES_9B = [  # Klein-9B (d 4096) and Dev (d 6144): N = d GEMMs are 288 / 432 tiles of 256^2
    (4608, 36864, 4096, "9b.linear1"),
    (4608, 4096, 16384, "9b.linear2"),
    (4608, 4096, 4096, "9b.proj"),
    (4608, 4096, 12288, "9b.mlp_out"),
    (4608, 6144, 24576, "dev.linear2"),
    (4608, 6144, 6144, "dev.proj"),
]


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


def main():
    ctx = Context(0)
    check = "--check" in sys.argv
    rounds = 3
    print(f"FLUX2MI_GEMM={os.environ.get('FLUX2MI_GEMM', '(auto)')}")
    for M, N, K, name in (SHAPES_9B if "--9b" in sys.argv else SHAPES):
        a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
        w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
        wt = w.t()
        c16 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        ct = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        t32, t16, tt = [], [], []
        for _ in range(rounds):   # interleaved, so that a clock / thermal drift hits all three alike
            with torch.cuda.stream(ctx.stream):
                t32.append(time_fn(lambda: ctx.op_gemm_bf16(a, w)))
                t16.append(time_fn(lambda: ctx.op_gemm_bf16_out_bf16(a, w, c16)))
            tt.append(time_fn(lambda: torch.matmul(a, wt, out=ct)))
        ms, ms16, ms_t = sorted(t32)[rounds // 2], sorted(t16)[rounds // 2], sorted(tt)[rounds // 2]
        fl = 2.0 * M * N * K
        if "--fp8" in sys.argv:   # GEMM kernel alone (HIP events inside the library), quantisation passes excluded
            for _ in range(3):
                ctx.op_gemm_fp8(a, w)
            ctx.prof_begin(1)
            for _ in range(10):
                ctx.op_gemm_fp8(a, w)
            n, pms, _ = ctx.prof_end()
            ms8 = pms / n
            print(f"{name:16s} fp8 e4m3 kernel {ms8:7.3f} ms {fl / ms8 / 1e9:7.1f} TF", flush=True)
        line = (f"{name:16s} M{M} N{N} K{K}: ours f32 C {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF | ours bf16 C {ms16:7.3f} ms {fl / ms16 / 1e9:7.1f} TF"
                f" | hipBLASLt bf16 C {ms_t:7.3f} ms {fl / ms_t / 1e9:7.1f} TF | ours/vendor {ms_t / ms16:5.3f}")
        if check:
            out = ctx.op_gemm_bf16(a, w)
            ref = torch.matmul(a.float(), w.float().t())
            line += f" | rel {float((out - ref).norm() / ref.norm()):.2e}"
            line += f" bf16-out == round(f32-out): {bool(torch.equal(ctx.op_gemm_bf16_out_bf16(a, w), out.to(torch.bfloat16)))}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
