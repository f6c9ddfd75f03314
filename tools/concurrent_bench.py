#!/usr/bin/env python3
"""Throughput experiment: N generations in flight on ONE GPU (N contexts = N HIP streams, N weight
replicas, N host threads).  Kernels of different images fill the CUs that a partial last round of
another image's kernel leaves idle.  Prints images/s and denoise-steps/s for N = 1, 2.  Dev tool."""
import os, sys, time, threading, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi import synthetic as syn
from flux2mi.pipeline import Flux2Pipeline

model = sys.argv[1] if len(sys.argv) > 1 else "klein-4b"
dev = torch.device("cuda", 0)
cfg, vcfg = syn.DIT_PRESETS[model], syn.VAE_PRESETS["standard"]
P = syn.make_dit_weights(cfg, device=dev)
VP = syn.make_vae_weights(vcfg, device=dev)
inp = syn.make_inputs(cfg, 1024, 1024, device=dev)
T = 28
pipes = []
for n in (1, 2):
    while len(pipes) < n:
        pipes.append(Flux2Pipeline(model=model, vae_variant="standard").load_models(P, VP))

    def work(p, reps):
        for _ in range(reps):
            p.generate_text_to_image(height=1024, width=1024, steps=T, precomputed_embeddings=inp["txt"], noise=inp["noise"])

    for p in pipes[:n]:
        work(p, 1)   # warm-up
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(p, reps)) for p in pipes[:n]]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"in_flight": n, "images_per_s": round(n * reps / dt, 3), "denoise_steps_per_s": round(n * reps * T / dt, 2),
                      "s_per_image_latency": round(dt / reps, 3)}), flush=True)
