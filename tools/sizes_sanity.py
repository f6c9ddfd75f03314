import sys, time, torch
sys.path.insert(0, "/root/repo")
import flux2mi
from flux2mi import synthetic as syn
from flux2mi.pipeline import Flux2Pipeline
dev = torch.device("cuda", 0)
cfg = syn.DIT_PRESETS["klein-4b"]
pipe = Flux2Pipeline(model="klein-4b", vae_variant="small").load_models(syn.make_dit_weights(cfg, device=dev), syn.make_vae_weights(syn.VAE_PRESETS["small"], device=dev))
for (h, w, steps) in [(2048, 2048, 4), (1000, 720, 4), (1536, 1024, 4), (64, 64, 2)]:
    inp = syn.make_inputs(cfg, *flux2mi.api.validate_dimensions(h, w), device=dev)
    t0 = time.perf_counter()
    res = pipe.generate_text_to_image(height=h, width=w, steps=steps, precomputed_embeddings=inp["txt"], noise=inp["noise"])
    torch.cuda.synchronize()
    im = torch.from_numpy(res.image)
    print(h, w, "->", tuple(im.shape), f"{time.perf_counter()-t0:.2f}s ms/step={res.msDenoise/steps:.1f} decode={res.msDecode:.1f}ms mean={im.float().mean():.1f} std={im.float().std():.1f} mem={torch.cuda.max_memory_allocated()/1e9:.1f}GB(torch)", flush=True)
