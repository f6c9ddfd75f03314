#!/usr/bin/env python3
"""One Klein-4B DiT forward at 1024x1024 (S = 4608), synthetic weights: the smallest program
that launches every hot kernel at its production shape.  Used under `rocprofv3 --pmc ...`
(counter passes serialise kernels, so the whole bench would take too long).  Dev tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi import synthetic as syn, api
from flux2mi.api import Context, DiT

model = sys.argv[1] if len(sys.argv) > 1 else "klein-4b"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
weights = sys.argv[3] if len(sys.argv) > 3 else "bf16"     # "fp8": weight_format 1
dev = torch.device("cuda", 0)
cfg = syn.DIT_PRESETS[model]
ctx = Context(0)
dit = DiT(ctx, model, weights=weights).load(syn.make_dit_weights(cfg, device=dev)).finalize()
inp = syn.make_inputs(cfg, 1024, 1024, device=dev)
lat = torch.randn((4096, 128), device=dev)
img_ids = torch.from_numpy(api.image_position_ids(1024, 1024)).to(dev)
txt_ids = torch.from_numpy(api.text_position_ids(512)).to(dev)
for _ in range(n):
    out = dit.forward(lat, inp["txt"], 0.5, 4.0, img_ids, txt_ids)
ctx.sync()
print("ok", float(out.abs().mean()))
