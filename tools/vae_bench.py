#!/usr/bin/env python3
"""VAE decode alone at 1024x1024 (latent 128x128), synthetic weights: ms per decode; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.  Dev tool."""
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
z = torch.randn((1, 32, 128, 128), device=dev)
for _ in range(2):
    vae.decode(z, want_f32=False)
ctx.sync()
t0 = time.perf_counter()
for _ in range(n):
    _, u8 = vae.decode(z, want_f32=False)
ctx.sync()
print(f"vae decode {variant} 1024x1024: {(time.perf_counter() - t0) / n * 1e3:.2f} ms; mean pixel {float(u8.float().mean()):.2f}")
