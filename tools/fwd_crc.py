import os, sys, zlib
ROOT = "/root/repo" if os.path.exists("/root/repo/flux2mi.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
import torch
import flux2mi
from flux2mi import synthetic as syn, api
from flux2mi.api import Context, DiT
dev = torch.device("cuda", 0)
cfg = syn.DIT_PRESETS["klein-4b"]
ctx = Context(0)
dit = DiT(ctx, "klein-4b", weights="bf16").load(syn.make_dit_weights(cfg, device=dev)).finalize()
inp = syn.make_inputs(cfg, 1024, 1024, device=dev)
g = torch.Generator(device=dev); g.manual_seed(3)
lat = torch.randn((4096, 128), device=dev, generator=g)
img_ids = torch.from_numpy(api.image_position_ids(1024, 1024)).to(dev)
txt_ids = torch.from_numpy(api.text_position_ids(512)).to(dev)
out = dit.forward(lat, inp["txt"], 0.5, 4.0, img_ids, txt_ids)
ctx.sync()
o = out.float().cpu()
print(os.environ.get("FLUX2MI_LIB", "libflux2mi.so"), "crc", "%08x" % (zlib.crc32(o.numpy().tobytes()) & 0xffffffff), "mean|x|", float(o.abs().mean()))
torch.save(o, f"{ROOT}/gpurun_out/fwd_{os.environ.get('FLUX2MI_LIB', 'libflux2mi.so')}.pt")
