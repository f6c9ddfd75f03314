#!/usr/bin/env python3
"""The stated per-pixel tolerance of the path (north_star: "outputs match ... within a stated per-pixel fp tolerance"):
a full 28-step Klein-4B generation + standard VAE decode, HIP path vs the fp32 CPU oracle on identical synthetic bf16
weights and inputs.  Prints per-step latent rel-L2 and the u8 image statistics as one JSON object.

  python tools/trajectory_tolerance.py 28 256                  both sides in one process on a GPU box (~3 min of oracle)
  python tools/trajectory_tolerance.py 28 1024 --oracle-only F the oracle side alone, on any CPU (no GPU, no library): 28 forwards
                                                               at S 4608 = 28 x ~1-2 min; per-step latents + image -> F (torch.save)
  python tools/trajectory_tolerance.py 28 1024 --against F     the HIP side on a GPU box against that file
  python tools/trajectory_tolerance.py 28 256 --fp8            weight_format 1 against the oracle running the same e4m3 emulation, and both
                                                               against the f32 oracle (profiles/r03_i_trajectory_tolerance_fp8_256.json)

The split form exists because BASELINE config B itself (1024 x 1024, S 4608) costs ~30-50 min of CPU oracle time — more than one GPU-box
call allows; weights and inputs are seeded on the CPU in all three forms, so both sides see bit-identical bf16 parameters.
Results: profiles/r02_k_trajectory_tolerance.json (256^2), profiles/r03_trajectory_tolerance_1024.json (config B), DESIGN.md §1c."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

argv = [a for a in sys.argv[1:]]
oracle_only = argv[argv.index("--oracle-only") + 1] if "--oracle-only" in argv else None
against = argv[argv.index("--against") + 1] if "--against" in argv else None
threads = int(argv[argv.index("--threads") + 1]) if "--threads" in argv else None
fp8 = "--fp8" in argv   # weight_format 1 against the oracle running the same e4m3 emulation (block GEMMs + Q K^T), one process only
pos = [a for i, a in enumerate(argv) if not a.startswith("--") and (i == 0 or argv[i - 1] not in ("--oracle-only", "--against", "--threads"))]
steps = int(pos[0]) if len(pos) > 0 else 28
H = W = int(pos[1]) if len(pos) > 1 else 256

from tests.conftest import host_threads

torch.set_num_threads(threads or host_threads())
import importlib.util

spec = importlib.util.spec_from_file_location("f2syn", os.path.join(ROOT, "flux-2-swift-mlx_amd", "synthetic.py"))
syn = importlib.util.module_from_spec(spec)
spec.loader.exec_module(syn)       # the weight recipe alone: the oracle-only form must not need libflux2mi.so
cfg, vcfg = syn.DIT_PRESETS["klein-4b"], syn.VAE_PRESETS["standard"]
P = syn.make_dit_weights(cfg)                         # CPU generator: the same values wherever this runs
VP = syn.make_vae_weights(vcfg, encoder=False)
inp = syn.make_inputs(cfg, H, W, s_txt=512)

if oracle_only:
    from oracle import flux2_ref as R
    t0 = time.time()
    marks = []

    def tick(c, x):
        marks.append(time.time() - t0)
        print(f"oracle step {len(marks)}/{steps}: {marks[-1]:.0f} s", file=sys.stderr, flush=True)
        return x
    with torch.no_grad():
        ref = R.generate(P, cfg, VP, vcfg, inp["txt"].float(), inp["noise"], H, W, steps, on_step=tick)
    torch.save({"latents": [l.to(torch.float32) for l in ref["latents"]], "image": ref["image"], "steps": steps, "H": H, "W": W,
                "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": torch.get_num_threads()}, oracle_only)
    print(json.dumps({"saved": oracle_only, "oracle_seconds": round(time.time() - t0, 1)}))
    sys.exit(0)

import flux2mi  # noqa: E402,F401
from flux2mi.pipeline import Flux2Pipeline  # noqa: E402

pipe = Flux2Pipeline(model="klein-4b", vae_variant="standard", quantization="fp8" if fp8 else "bf16").load_models(P, VP)
traj = []
res = pipe.generate_text_to_image(height=H, width=W, steps=steps, precomputed_embeddings=inp["txt"], noise=inp["noise"],
                                  on_step=lambda c, x: traj.append(x.detach().clone().cpu()) or x)
if against:
    ref = torch.load(against)
    assert ref["steps"] == steps and ref["H"] == H, (ref["steps"], ref["H"])
    osec, othr = ref["oracle_seconds"], ref["oracle_threads"]
else:
    from oracle import flux2_ref as R
    import contextlib
    t0 = time.time()

    def tick(c, x):
        print(f"oracle step at {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
        return x
    with torch.no_grad(), (R.fp8_block_gemms(device_roundings=True) if fp8 else contextlib.nullcontext()):   # the emulation with the HIP path's bf16 store points
        ref = R.generate(P, cfg, VP, vcfg, inp["txt"].float(), inp["noise"], H, W, steps, on_step=tick)
    ref32 = None
    if fp8:   # the unquantised oracle as well: how large the quantisation effect itself is on this trajectory
        with torch.no_grad():
            ref32 = R.generate(P, cfg, VP, vcfg, inp["txt"].float(), inp["noise"], H, W, steps, on_step=tick)
    osec, othr = round(time.time() - t0, 1), torch.get_num_threads()
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
errs = [rel(traj[k], ref["latents"][k]) for k in range(steps)]
d = (torch.from_numpy(res.image).int() - ref["image"].int()).abs().float()
out = {"workload": f"Klein-4B full depth, {H}x{W}, {steps} steps, standard VAE decode; HIP path vs fp32 CPU oracle, synthetic bf16 weights (CPU-seeded)",
       "latent_rel_l2_per_step": [round(e, 6) for e in errs], "latent_rel_l2_final": round(errs[-1], 6), "latent_rel_l2_max": round(max(errs), 6),
       "u8_abs_diff": {"mean": round(float(d.mean()), 4), "max": int(d.max()), "frac_gt_1": round(float((d > 1).float().mean()), 5),
                       "frac_gt_2": round(float((d > 2).float().mean()), 5), "frac_gt_8": round(float((d > 8).float().mean()), 6)},
       "oracle_seconds": osec, "oracle_threads": othr, "oracle_side": "separate CPU run (--oracle-only)" if against else "same process"}
if fp8:
    q = [rel(ref["latents"][k], ref32["latents"][k]) for k in range(steps)]
    d32 = (torch.from_numpy(res.image).int() - ref32["image"].int()).abs().float()
    dq = (ref["image"].int() - ref32["image"].int()).abs().float()
    out["workload"] = out["workload"].replace("HIP path vs fp32 CPU oracle", "HIP fp8 path (e4m3 block GEMMs + e4m3 Q K^T) vs the CPU oracle running the same e4m3 emulation")
    out["quantisation_effect_emulating_vs_f32_oracle"] = {"latent_rel_l2_final": round(q[-1], 6), "latent_rel_l2_max": round(max(q), 6), "u8_mean": round(float(dq.mean()), 4), "u8_max": int(dq.max())}
    out["hip_fp8_vs_f32_oracle"] = {"latent_rel_l2_final": round(rel(traj[-1], ref32["latents"][-1]), 6), "u8_mean": round(float(d32.mean()), 4), "u8_max": int(d32.max())}
print(json.dumps(out))
