#!/usr/bin/env python3
"""VAE decode alone at 1024x1024 (latent 128x128; `vae_bench.py VARIANT N H W` for another latent size), synthetic weights: ms per
decode, the mid block's attention by HIP events (kernel class 8), peak library memory; run it under `rocprofv3 --kernel-trace --stats`
for the per-kernel split.  Dev tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi import synthetic as syn
from flux2mi.api import Context, VAE

variant = sys.argv[1] if len(sys.argv) > 1 else "standard"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
ctx = Context(0)
vcfg = syn.VAE_PRESETS[variant]
vae = VAE(ctx, vcfg).load(syn.make_vae_weights(vcfg, device=dev, encoder=False)).finalize()
lh = int(sys.argv[3]) if len(sys.argv) > 3 else 128
lw = int(sys.argv[4]) if len(sys.argv) > 4 else 128
z = torch.randn((1, 32, lh, lw), device=dev)
for _ in range(2):
    vae.decode(z, want_f32=False)
ctx.sync()
t0 = time.perf_counter()
for _ in range(n):
    _, u8 = vae.decode(z, want_f32=False)
ctx.sync()
ms = (time.perf_counter() - t0) / n * 1e3
ctx.prof_begin(8)
vae.decode(z, want_f32=False)
na, ms_a, fl = ctx.prof_end()
peak = ctx.memory_stats()["peak"] / float(1 << 30)
print(f"vae decode {variant} {8 * lw}x{8 * lh}: {ms:.2f} ms; mid attention {ms_a:.3f} ms ({fl / max(ms_a, 1e-9) / 1e9:.0f} TFLOP/s on 4 P^2 C); "
      f"peak library memory {peak:.2f} GiB; mean pixel {float(u8.float().mean()):.2f}")
