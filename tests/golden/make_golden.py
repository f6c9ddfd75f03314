#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the oracle (oracle/flux2_ref.py) on synthetic
seeded weights.  The reference is Swift/MLX and cannot be imported or run here
(SURVEY.md §8c), so these vectors pin the ORACLE (drift detection + a second check
for the HIP path), not the reference.  Usage: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import flux2mi  # noqa: E402
from flux2mi import synthetic as syn  # noqa: E402
from oracle import flux2_ref as R  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
PROD_DIT_SLICE = dict(dim=3072, heads=24, n_double=1, n_single=1, joint_dim=7680, guidance=False)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    # ---- scheduler known answers (cross-checked against the reference's formulas)
    tables = {}
    for name, (steps, seq, strength) in {"n28_s4096": (28, 4096, 1.0), "n50_s4096": (50, 4096, 1.0),
                                          "n50_s4096_str07": (50, 4096, 0.7), "n4_s256": (4, 256, 1.0),
                                          "n4_s4096": (4, 4096, 1.0), "n28_s8192": (28, 8192, 1.0)}.items():
        s, t0 = R.set_timesteps(steps, seq, strength)
        tables[name] = np.array(s, dtype=np.float32)
        tables[name + "_tstart"] = np.array([t0])
    np.savez(os.path.join(OUT, "scheduler_sigmas.npz"), **tables)

    # ---- tiny DiT forward + 4-step trajectory with RePaint hook and strength 0.7
    cfg, vcfg = syn.DIT_PRESETS["tiny"], syn.VAE_PRESETS["tiny"]
    P, VP = syn.make_dit_weights(cfg), syn.make_vae_weights(vcfg)
    inp = syn.make_inputs(cfg, 128, 128, s_txt=16)
    lat = R.pack_patchified_to_sequence(inp["noise"])
    taps = {}
    out = R.dit_forward(P, cfg, lat, inp["txt"].float(), torch.tensor([0.731]), None,
                        R.image_position_ids(128, 128), R.text_position_ids(16), taps=taps)
    gen = R.generate(P, cfg, VP, vcfg, inp["txt"].float(), inp["noise"], 128, 128, 4)
    g = torch.Generator().manual_seed(99)
    image_latents = torch.randn((1, 64, 128), generator=g)
    blend_noise = torch.randn((1, 64, 128), generator=g)
    mask = torch.zeros((8, 8))
    mask[2:6, 2:6] = 1.0
    mask = mask.reshape(1, 64, 1)
    hook = lambda c, x: R.repaint_blend(x, image_latents, blend_noise, mask, c["sigmaNext"])
    gen2 = R.generate(P, cfg, VP, vcfg, inp["txt"].float(), inp["noise"], 128, 128, 10, init_latents=image_latents,
                      strength=0.7, on_step=hook)
    np.savez_compressed(
        os.path.join(OUT, "tiny_dit.npz"),
        forward_out=out[0].numpy().astype(np.float32),
        temb=taps["temb"][0].numpy(), rope_cos=taps["cos"].numpy(), rope_sin=taps["sin"].numpy(),
        double0_img=taps["double0.img"][0].numpy(), single1=taps["single1"][0].numpy(),
        t2i_final_latents=gen["final_latents"][0].numpy(), t2i_image=gen["image"].numpy(),
        t2i_sigmas=np.array(gen["sigmas"], dtype=np.float32),
        repaint_final_latents=gen2["final_latents"][0].numpy(), repaint_image=gen2["image"].numpy(),
        repaint_sigmas=np.array(gen2["sigmas"], dtype=np.float32))

    # ---- tiny VAE decode
    gz = torch.Generator().manual_seed(11)
    z = torch.randn((1, 32, 8, 8), generator=gz)
    vt = {}
    dec = R.vae_decode(VP, vcfg, z, taps=vt)
    np.savez_compressed(os.path.join(OUT, "tiny_vae.npz"), decoded=dec[0].numpy().astype(np.float16),
                        image=R.postprocess_vae_output(dec).numpy(), mid=vt["mid"][0].numpy().astype(np.float16))
    # ---- production-WIDTH slices: Klein-4B widths (d 3072, 24 heads, J 7680) with one
    # double + one single block, and the standard VAE decoder widths [128,256,512,512] on an 8x8
    # latent.  Weights are regenerated from the seed by the tests (synthetic.py), only I/O is stored.
    pcfg = dict(PROD_DIT_SLICE)
    PP = syn.make_dit_weights(pcfg, seed=2024)
    pin = syn.make_inputs(pcfg, 128, 128, s_txt=32, noise_seed=5, txt_seed=6)
    plat = R.pack_patchified_to_sequence(pin["noise"])
    ptaps = {}
    pout = R.dit_forward(PP, pcfg, plat, pin["txt"].float(), torch.tensor([0.4]), None,
                         R.image_position_ids(128, 128), R.text_position_ids(32), taps=ptaps)
    np.savez_compressed(os.path.join(OUT, "prod_dit_slice.npz"), forward_out=pout[0].numpy().astype(np.float32),
                        double0_img=ptaps["double0.img"][0].numpy().astype(np.float16),
                        single0=ptaps["single0"][0].numpy().astype(np.float16))
    del PP
    svcfg = syn.VAE_PRESETS["standard"]
    SVP = syn.make_vae_weights(svcfg, seed=77)
    z = torch.randn((1, 32, 8, 8), generator=torch.Generator().manual_seed(12))
    sdec = R.vae_decode(SVP, svcfg, z)
    np.savez_compressed(os.path.join(OUT, "prod_vae.npz"), decoded=sdec[0].numpy().astype(np.float16),
                        image=R.postprocess_vae_output(sdec).numpy())
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
