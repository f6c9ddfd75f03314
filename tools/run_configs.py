#!/usr/bin/env python3
"""Runs the BASELINE.json configurations that are not the bench headline, once each, and prints
their timings (synthetic weights/inputs).  A: Klein-4B 256^2 4 steps; C: Dev 1024^2 50 steps;
D: Klein-9B 1024^2 28 steps, bf16 weights; D8: the same with e4m3 block GEMMs (weight_format 1);
E: Dev 1024^2, 50 steps at strength 0.7 (35 effective) with the RePaint hook, T2I and I2I (one
1024^2 reference image, S = 8704)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi import synthetic as syn, api
from flux2mi.pipeline import Flux2Pipeline, Flux2MaskedInpaintingChain

which = sys.argv[1:] or ["A", "D", "C", "E"]
dev = torch.device("cuda", 0)


def load(model, vae="standard", quantization="bf16"):
    pipe = Flux2Pipeline(model=model, vae_variant=vae, quantization=quantization)
    cfg = syn.DIT_PRESETS[model[:-3] if model.endswith("-kv") else model]
    pipe.load_models(syn.make_dit_weights(cfg, device=dev), syn.make_vae_weights(syn.VAE_PRESETS[vae], device=dev))
    return pipe, cfg


def run(name, fn):
    fn()  # warm-up (allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": name, "seconds_per_image": round(dt, 3), "effective_steps": res.effectiveSteps,
                      "ms_per_denoise_step": round(res.msDenoise / max(res.effectiveSteps, 1), 2),
                      "decode_ms": round(res.msDecode, 1)}), flush=True)


for w in which:
    if w == "A":
        pipe, cfg = load("klein-4b")
        inp = syn.make_inputs(cfg, 256, 256, device=dev)
        run("A klein-4b 256x256 4 steps", lambda: pipe.generate_text_to_image(height=256, width=256, steps=4, precomputed_embeddings=inp["txt"], noise=inp["noise"]))
    elif w == "D":
        pipe, cfg = load("klein-9b")
        inp = syn.make_inputs(cfg, 1024, 1024, device=dev)
        run("D klein-9b (bf16 weights) 1024x1024 28 steps, 1 image per GPU", lambda: pipe.generate_text_to_image(height=1024, width=1024, steps=28, precomputed_embeddings=inp["txt"], noise=inp["noise"]))
    elif w == "D8":
        pipe, cfg = load("klein-9b", quantization="fp8")
        inp = syn.make_inputs(cfg, 1024, 1024, device=dev)
        run("D8 klein-9b (fp8 e4m3 block GEMMs) 1024x1024 28 steps, 1 image per GPU", lambda: pipe.generate_text_to_image(height=1024, width=1024, steps=28, precomputed_embeddings=inp["txt"], noise=inp["noise"]))
    elif w == "K":
        # klein-9b-kv: one 1024^2 reference image (S_ref 4096), 4 steps like the distilled model; the
        # same I2I generation without and with the KV-cached path
        pipe, cfg = load("klein-9b-kv")
        inp = syn.make_inputs(cfg, 1024, 1024, device=dev)
        g = torch.Generator(device=dev); g.manual_seed(99)
        ref = torch.randn((1, 4096, 128), generator=g, device=dev)
        ref_ids = torch.from_numpy(api.reference_position_ids([64], [64])).to(dev)
        for kvc in (False, True):
            run(f"K klein-9b{'-kv (KV-cached reference tokens)' if kvc else ' plain I2I'} 1024x1024 + one 1024^2 reference, 4 steps",
                lambda: pipe.generate_with_result("imageToImage", height=1024, width=1024, steps=4, precomputed_embeddings=inp["txt"],
                                                  noise=inp["noise"], reference_latents=ref, reference_ids=ref_ids, kv_cache=kvc))
    elif w == "C":
        pipe, cfg = load("dev")
        inp = syn.make_inputs(cfg, 1024, 1024, device=dev)
        run("C dev 1024x1024 50 steps", lambda: pipe.generate_text_to_image(height=1024, width=1024, steps=50, guidance=4.0, precomputed_embeddings=inp["txt"], noise=inp["noise"]))
    elif w == "E":
        pipe, cfg = load("dev")
        inp = syn.make_inputs(cfg, 1024, 1024, device=dev)
        g = torch.Generator(device=dev); g.manual_seed(99)
        S = 4096
        img_lat = torch.randn((1, S, 128), generator=g, device=dev)
        blend = torch.randn((1, S, 128), generator=g, device=dev)
        m = torch.zeros((64, 64), device=dev); m[16:48, 16:48] = 1.0   # centred square, 25 % of the tokens
        chain = Flux2MaskedInpaintingChain(pipe, img_lat, m.reshape(1, S, 1), blend)
        run("E dev 1024x1024 strength 0.7 of 50 + RePaint hook (T2I path, S=4608)",
            lambda: chain.run(inp["txt"], 1024, 1024, steps=50, guidance=4.0, strength=0.7, noise=inp["noise"]))
        ref_ids = torch.from_numpy(api.reference_position_ids([64], [64])).to(dev)
        run("E dev 1024x1024 strength 0.7 of 50 + RePaint hook + one 1024^2 reference (I2I path, S=8704)",
            lambda: chain.run(inp["txt"], 1024, 1024, steps=50, guidance=4.0, strength=0.7, noise=inp["noise"],
                              reference_latents=img_lat, reference_ids=ref_ids))
    del pipe
    torch.cuda.empty_cache()
