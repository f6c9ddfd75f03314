#!/usr/bin/env python3
"""The stated per-pixel tolerance of the path (north_star: "outputs match ... within a stated per-pixel fp tolerance"):
a full 28-step Klein-4B generation at 256x256 + standard VAE decode, HIP path vs the fp32 CPU oracle on identical
synthetic bf16 weights and inputs.  Prints per-step latent rel-L2 and the u8 image statistics as one JSON object
(committed as profiles/r02_trajectory_tolerance.json, quoted in DESIGN.md §1c).  ~3 minutes of CPU oracle time."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import flux2mi
from flux2mi import synthetic as syn
from flux2mi.pipeline import Flux2Pipeline
from oracle import flux2_ref as R
from tests.conftest import host_threads

torch.set_num_threads(host_threads())
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 28
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg, vcfg = syn.DIT_PRESETS["klein-4b"], syn.VAE_PRESETS["standard"]
P = syn.make_dit_weights(cfg, device="cuda")
VP = syn.make_vae_weights(vcfg, encoder=False)
pipe = Flux2Pipeline(model="klein-4b", vae_variant="standard").load_models(P, VP)
Pc = {k: v.cpu() for k, v in P.items()}
del P
inp = syn.make_inputs(cfg, H, W, s_txt=512)
traj = []
res = pipe.generate_text_to_image(height=H, width=W, steps=steps, precomputed_embeddings=inp["txt"], noise=inp["noise"],
                                  on_step=lambda c, x: traj.append(x.detach().clone().cpu()) or x)
t0 = time.time()
with torch.no_grad():
    ref = R.generate(Pc, cfg, VP, vcfg, inp["txt"].float(), inp["noise"], H, W, steps)
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
errs = [rel(traj[k], ref["latents"][k]) for k in range(steps)]
d = (torch.from_numpy(res.image).int() - ref["image"].int()).abs().float()
out = {"workload": f"Klein-4B full depth, {H}x{W}, {steps} steps, standard VAE decode; HIP path vs fp32 CPU oracle, synthetic bf16 weights",
       "latent_rel_l2_per_step": [round(e, 6) for e in errs], "latent_rel_l2_final": round(errs[-1], 6), "latent_rel_l2_max": round(max(errs), 6),
       "u8_abs_diff": {"mean": round(float(d.mean()), 4), "max": int(d.max()), "frac_gt_1": round(float((d > 1).float().mean()), 5),
                       "frac_gt_2": round(float((d > 2).float().mean()), 5), "frac_gt_8": round(float((d > 8).float().mean()), 6)},
       "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": host_threads()}
print(json.dumps(out))
