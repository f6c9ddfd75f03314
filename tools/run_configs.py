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
    elif w == "T":
        # Klein-4B prompt embeddings: Qwen3-4B, 27 of 36 layers, 512 tokens (synthetic weights made on the GPU)
        from flux2mi.textenc import TextEncoder
        ctx = api.Context(0)
        te = TextEncoder(ctx, "klein-4b")
        g = torch.Generator(device=dev); g.manual_seed(3)
        c = te.cfg
        d, I, nq, nkv = c.hidden, c.intermediate, c.heads * 128, c.kv_heads * 128
        rnd = lambda *sh, s=1.0: (torch.randn(sh, generator=g, device=dev) * s).to(torch.bfloat16)
        P = {"embed_tokens.weight": rnd(c.vocab, d)}
        for l in range(27):
            p_ = f"layers.{l}."
            P.update({p_ + "self_attn.q_proj.weight": rnd(nq, d, s=d ** -0.5), p_ + "self_attn.k_proj.weight": rnd(nkv, d, s=d ** -0.5),
                      p_ + "self_attn.v_proj.weight": rnd(nkv, d, s=d ** -0.5), p_ + "self_attn.o_proj.weight": rnd(d, nq, s=nq ** -0.5),
                      p_ + "self_attn.q_norm.weight": 1 + 0.1 * rnd(128), p_ + "self_attn.k_norm.weight": 1 + 0.1 * rnd(128),
                      p_ + "mlp.gate_proj.weight": rnd(I, d, s=d ** -0.5), p_ + "mlp.up_proj.weight": rnd(I, d, s=d ** -0.5),
                      p_ + "mlp.down_proj.weight": rnd(d, I, s=I ** -0.5), p_ + "input_layernorm.weight": 1 + 0.1 * rnd(d),
                      p_ + "post_attention_layernorm.weight": 1 + 0.1 * rnd(d)})
        te.load(P).finalize()
        del P
        ids = torch.randint(0, 150000, (512,), generator=g, device=dev).to(torch.int32)
        for valid in (40, 512):
            te.forward(ids, valid); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                out = te.forward(ids, valid)
            ctx.sync()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            fl = 27 * 2.0 * 512 * (d * (nq + 2 * nkv) + nq * d + 3 * d * I)
            print(json.dumps({"config": f"T klein-4b prompt embeddings (Qwen3-4B, 27 layers, S 512, {valid} real tokens)",
                              "ms": round(ms, 2), "linear_tflops": round(fl / ms / 1e9, 1), "param_GB": round(te.param_bytes / 1e9, 2)}), flush=True)
        te.close()
        continue
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
