#!/usr/bin/env python3
"""Dev tool for the 4-wave GEMM (gemm256g4.hip): correctness on edge shapes, then timing against the 8-wave kernel and
hipBLASLt, interleaved in one process.  Run with FLUX2MI_LIB=libflux2mi_dev.so; this script flips FLUX2MI_GEMM_V itself
by re-loading nothing: the dev library reads the switch once, so each variant is a separate process (--variant)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(4608, 27648, 3072, "linear1"), (4608, 3072, 12288, "linear2"), (4608, 3072, 9216, "mlp_out"), (4608, 9216, 3072, "qkv"),
          (4608, 3072, 3072, "proj"), (4608, 18432, 3072, "mlp_in"), (8192, 8192, 8192, "square8k"), (4096, 4096, 4096, "square4k")]


BF16 = "--bf16" in sys.argv


def child(variant):
    import torch
    import flux2mi
    from flux2mi.api import Context
    ctx = Context(0)
    if "--check" in sys.argv:
        g = torch.Generator().manual_seed(0)
        for M, N, K in [(256, 256, 64), (256, 256, 128), (256, 256, 192), (256, 256, 256), (256, 256, 320), (300, 500, 448), (1000, 777 + 7, 1024),
                        (4608, 3072, 3072), (513, 256, 2048), (255, 1024, 64)]:
            a = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
            w = (torch.randn((N, K), generator=g) * 0.05).to(torch.bfloat16).cuda()
            out = ctx.op_gemm_bf16_out_bf16(a, w).float() if BF16 else ctx.op_gemm_bf16(a, w)
            ref = a.float() @ w.float().t()
            if BF16:
                ref = ref.to(torch.bfloat16).float()
            err = float((out - ref).norm() / ref.norm())
            bad = int((~torch.isfinite(out)).sum())
            print(f"check {variant} M{M} N{N} K{K}: rel {err:.2e} nonfinite {bad}", flush=True)
            assert err < 2e-3 and bad == 0
        # exact-data race screen: integer operands, repeated
        a = torch.randint(-3, 4, (4608, 3072)).to(torch.bfloat16).cuda()
        w = torch.randint(-3, 4, (3072, 3072)).to(torch.bfloat16).cuda()
        ref = a.float() @ w.float().t()
        for it in range(20):
            assert torch.equal(ctx.op_gemm_bf16_out_bf16(a, w).float() if BF16 else ctx.op_gemm_bf16(a, w), ref.to(torch.bfloat16).float() if BF16 else ref), it
        print(f"check {variant}: exact-data x20 ok", flush=True)
    for M, N, K, name in SHAPES:
        a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
        w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
        c16 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        wt = w.t()
        fn = (lambda: torch.matmul(a, wt, out=c16)) if variant == "blaslt" else ((lambda: ctx.op_gemm_bf16_out_bf16(a, w, c16)) if BF16 else (lambda: ctx.op_gemm_bf16(a, w)))
        ts = []
        for r in range(5):
            with torch.cuda.stream(ctx.stream):
                for _ in range(3):
                    fn()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    fn()
                e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 10)
        med = sorted(ts)[2]
        print(f"{variant:7s} {name:9s} {med:7.3f} ms {2.0 * M * N * K / med / 1e9:7.0f} TF", flush=True)


if __name__ == "__main__":
    if "--variant" in sys.argv:
        child(sys.argv[sys.argv.index("--variant") + 1])
    else:
        for v, env in (("v2-8wave", {"FLUX2MI_GEMM_V": "2"}), ("v6-g4", {"FLUX2MI_GEMM_V": "6"}), ("blaslt", {})):
            e = dict(os.environ, FLUX2MI_LIB="libflux2mi_dev.so", FLUX2MI_GEMM="256", **env)
            subprocess.run([sys.executable, __file__, "--variant", v] + [x for x in sys.argv[1:] if x in ("--check", "--bf16")], env=e, check=False)
