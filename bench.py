#!/usr/bin/env python3
"""bench.py — Flux.2 Klein-4B 1024x1024 @ 28 steps on N MI355X (BASELINE.json configs[1]); `--config A..E` selects the other
BASELINE.json configurations under the same JSON contract (C: Dev 50 steps; D: Klein-9B fp8; E: Dev at strength 0.7 with the RePaint
step hook, `--reference` for its I2I form), e.g. `--gpus 8 --config D`, `--gpus 4 --config E`.

One bench "step" = one batch of N images (one per GPU, batch-level data parallel): the
28 denoise steps of the DiT, VAE decode, the all-gather of the decoded u8 images and the D2H
copy of the batch on rank 0, with text embeddings and noise resident in HBM when the timed
region starts.  Per batch the ranks exchange exactly two collectives (RCCL over xGMI, inside libflux2mi.so on the
stream the pipeline generates on: flux2mi_dp_broadcast_embeddings / flux2mi_dp_all_gather_images):
broadcast of the text embeddings from rank 0 and all-gather of the decoded images (device
buffers, no host staging).  torch.distributed is only the control plane (a gloo group on 127.0.0.1 that
carries the RCCL unique id and the CPU-side barrier).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself (fresh child processes, spawned before anything in this process touches a GPU); under
`torch.distributed.run` (WORLD_SIZE set) it is one of the ranks.  WORLD_SIZE != --gpus is an error.

Prints ONE JSON line on rank 0 (contract in the task statement) with
  value     = total denoise steps of all ranks / wall time of the timed region (incl. decode)
  roofline  = dominant kernel (single-stream linear1 GEMM) measured with HIP events on the
              library's stream during the timed region, vs the 2.5 PF dense bf16 MFMA peak
  cpu_baseline = the restated fp32 oracle (NOT MLX) timed on the host cores on ONE denoise step
              of this very workload (Klein-4B, S = 4608), extrapolated x28 for the image.
Weights are synthetic (seeded N(0,0.02), bf16) — there are no checkpoints on the box.
`--dry-run` exercises launcher + rendezvous + both collectives over gloo on CPU tensors with the
hot path replaced by a placeholder (no compute, no library): a harness self-test, never a result.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# BASELINE.json `configs`, by letter (A = configs[0] ... E = configs[4]).  `--gpus` is the caller's: D is quoted on 8, E on 4.
CONFIGS = {
    "A": {"name": "configs[0]: Klein-4B bf16, 256x256, 4 steps, batch 1 (the reference's CPU-runnable plumbing case, here on the GPU)",
          "model": "klein-4b", "weights": "bf16", "height": 256, "width": 256, "denoise_steps": 4, "strength": 1.0, "repaint": False},
    "B": {"name": "configs[1]: Klein-4B bf16, 1024x1024, 28 steps, batch 1 per GPU (the headline)",
          "model": "klein-4b", "weights": "bf16", "height": 1024, "width": 1024, "denoise_steps": 28, "strength": 1.0, "repaint": False},
    "C": {"name": "configs[2]: Dev (32B) bf16, 1024x1024, 50 steps, batch 1 per GPU",
          "model": "dev", "weights": "bf16", "height": 1024, "width": 1024, "denoise_steps": 50, "strength": 1.0, "repaint": False},
    "D": {"name": "configs[3]: Klein-9B fp8 weights, 1024x1024, 28 steps, one image per GPU (quoted at batch 8 on 8 GPUs)",
          "model": "klein-9b", "weights": "fp8", "height": 1024, "width": 1024, "denoise_steps": 28, "strength": 1.0, "repaint": False},
    "E": {"name": "configs[4]: Dev image-to-image at strength 0.7 with the Flux2MaskedInpaintingChain RePaint step hook, 1024x1024, "
                  "one image per GPU (quoted at batch 4 on 4 GPUs)",
          "model": "dev", "weights": "bf16", "height": 1024, "width": 1024, "denoise_steps": 50, "strength": 0.7, "repaint": True},
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed batches (one image per GPU each)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="B", choices=sorted(CONFIGS),
                    help="BASELINE.json configuration (default B = configs[1], the headline); sets the defaults of --model / --weights / "
                         "--height / --width / --denoise-steps / --strength / --repaint, each of which may still be given explicitly")
    ap.add_argument("--model", default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--denoise-steps", type=int, default=None)
    ap.add_argument("--vae", default="standard", choices=["standard", "small"])
    ap.add_argument("--weights", default=None, choices=["bf16", "fp8"],
                    help="transformer block-GEMM format; the headline (BASELINE.json configs[1]) is bf16")
    ap.add_argument("--strength", type=float, default=None, help="< 1: image-to-image start from seeded init latents (config E: 0.7)")
    ap.add_argument("--repaint", action="store_true", default=None,
                    help="run the Flux2MaskedInpaintingChain RePaint step hook every step (config E)")
    ap.add_argument("--reference", action="store_true",
                    help="config E's I2I form: one 1024^2 reference image (4096 reference tokens behind the output tokens, S = 8704)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the extra untimed generations that fill roofline_secondary / roofline_hbm (profiler runs)")
    ap.add_argument("--attn-softmax", default="auto", choices=["auto", "online"],
                    help="auto: fixed zero softmax offset where the QK-RMSNorm weights bound the scores, a per-query data "
                         "offset otherwise (what a checkpoint with larger norm scales gets: exact for any scores); "
                         "online: force the data-offset form in every block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mfma-stream", action="store_true",
                    help="skip the register-only MFMA stream after the timed region (roofline.mfma_stream): profiler runs")
    ap.add_argument("--prof-class", type=int, default=0, help="kernel class timed with HIP events (0 = linear1 GEMM)")
    ap.add_argument("--dry-run", action="store_true", help="harness self-test on CPU/gloo: no GPU, no compute")
    args = ap.parse_args(argv)
    for k, v in CONFIGS[args.config].items():
        if k != "name" and getattr(args, k) is None:
            setattr(args, k, v)
    return args


def launch_ranks(args) -> int:
    """Start --gpus fresh rank processes of this script (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* as torchrun sets them)
    and wait for them.  Nothing here imports torch or makes a HIP call: the children are the first to touch a GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:          # a rank died: the others would hang in a collective
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def host_threads() -> int:
    """Threads for the CPU baseline: the cgroup CPU quota where there is one (x2: measured best), else the visible CPUs."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, 2 * int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(launch_ranks(args))
    if int(env_world or "1") != max(1, args.gpus):
        sys.exit(f"bench.py: WORLD_SIZE={env_world} but --gpus {args.gpus}: start it as `python bench.py --gpus N` "
                 f"(self-launching) or under torch.distributed.run with --nproc-per-node equal to --gpus")

    import torch

    import flux2mi
    from flux2mi import parallel as par
    from flux2mi import synthetic as syn

    dry = args.dry_run
    rank, world, local = par.init_distributed()
    assert world == max(1, args.gpus), (world, args.gpus)
    if dry:
        dev = torch.device("cpu")
    else:
        dev_index = local if world > 1 else 0
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)

    cfg = syn.DIT_PRESETS[args.model]
    vcfg = syn.VAE_PRESETS[args.vae]
    H, W, T = args.height, args.width, args.denoise_steps
    s_txt = 512
    s_img = (H // 16) * (W // 16)
    S = s_txt + s_img
    d = cfg["dim"]

    # ---- model: synthetic weights generated on the GPU, replicated per rank
    t0 = time.time()
    pipe = P = None
    if not dry:
        from flux2mi.pipeline import Flux2Pipeline
        pipe = Flux2Pipeline(model=args.model, vae_variant=args.vae, device=dev.index, quantization=args.weights)
        # per-tensor seeds, generated on the GPU one tensor at a time (Dev is 64 GB: no second copy of the model anywhere)
        # the CPU-oracle leg needs the tensors themselves: Klein-4B only (a Dev forward is minutes of host time and 64 GB of host memory)
        need_host_copy = rank == 0 and not args.no_cpu_baseline and world == 1 and args.model == "klein-4b"
        # ONE weight recipe in every mode (default run, --no-cpu-baseline, other models, multi-GPU): output_check.image_crc32 of a
        # configuration is the same number whichever way it was run; the oracle leg draws the same tensors again (to the host)
        P = syn.LazyDitWeights(cfg, seed=1234, device=dev)
        VP = syn.make_vae_weights(vcfg, seed=4321, device=dev)
        pipe.load_models(P, VP)
        if world > 1:
            par.attach_context(pipe.ctx)   # this rank's RCCL communicator, on the stream the pipeline generates on
        if args.attn_softmax == "online":
            pipe.transformer.set_attention_softmax("online")
        if not need_host_copy:
            del P
            P = None
        del VP
        torch.cuda.synchronize()
    t_load = time.time() - t0

    # ---- inputs resident in HBM: per-rank noise (seed 42 + rank), text embeddings on rank 0
    gen = torch.Generator(device=dev)
    gen.manual_seed(42 + rank)
    noise = torch.randn((1, 128, H // 16, W // 16), generator=gen, device=dev, dtype=torch.float32)
    J = cfg["joint_dim"] if not dry else 64
    emb_shape = (world, s_txt, J)
    emb_all = None
    if rank == 0:
        g7 = torch.Generator(device=dev)
        g7.manual_seed(7)
        emb_all = torch.randn(emb_shape, generator=g7, device=dev, dtype=torch.float32).to(torch.bfloat16)
    guidance = 4.0
    # config E (Flux2MaskedInpaintingChain.swift:380-431): VAE-encoded source latents (seed 99 + rank), blend noise drawn once, a
    # centred square mask over 25 % of the tokens; strength < 1 starts from the source latents; --reference appends them as one reference image
    strength = float(args.strength)
    chain = ref_lat = ref_ids = init_lat = None
    if (args.repaint or strength < 1.0 or args.reference) and not dry:
        from flux2mi import api
        from flux2mi.pipeline import Flux2MaskedInpaintingChain
        g99 = torch.Generator(device=dev)
        g99.manual_seed(99 + rank)
        src_lat = torch.randn((1, s_img, 128), generator=g99, device=dev)
        if strength < 1.0:
            init_lat = src_lat
        if args.repaint:
            blend = torch.randn((1, s_img, 128), generator=g99, device=dev)
            hh, ww = H // 16, W // 16
            mask = torch.zeros((hh, ww), device=dev)
            mask[hh // 4: hh - hh // 4, ww // 4: ww - ww // 4] = 1.0
            chain = Flux2MaskedInpaintingChain(pipe, src_lat, mask.reshape(1, s_img, 1), blend)
        if args.reference:
            ref_lat = src_lat
            ref_ids = torch.from_numpy(api.reference_position_ids([H // 16], [W // 16])).to(dev)
    hook_calls = [0]

    def on_step(ctx_, latents):
        hook_calls[0] += 1
        return chain.on_step(ctx_, latents)
    img_dev = torch.zeros((1, H, W, 3), dtype=torch.uint8, device=dev)      # this rank's decoded image, in HBM
    host_batch = torch.empty((world, H, W, 3), dtype=torch.uint8, pin_memory=not dry) if rank == 0 else None

    def one_batch():
        emb = par.broadcast_text_embeddings(emb_all, emb_shape, dev, src=0)          # collective 1
        if dry:
            res = None
            img_dev.fill_(rank + 1)                                                  # placeholder, no compute
        else:
            res = pipe.generate_with_result("imageToImage" if ref_lat is not None else "textToImage", height=H, width=W, steps=T,
                                            guidance=guidance, precomputed_embeddings=emb[rank], noise=noise, image_device=img_dev[0],
                                            return_latents=True, init_latents=init_lat, strength=strength,
                                            on_step=on_step if chain is not None else None,
                                            reference_latents=ref_lat, reference_ids=ref_ids)
        imgs = par.all_gather_images(img_dev)                                        # collective 2 (device buffers)
        if rank == 0:
            host_batch.copy_(imgs, non_blocking=False)                               # the batch's D2H copy
        return res

    for _ in range(args.warmup):
        one_batch()

    par.barrier()
    sampler = None
    if world > 1:
        par.collective_stats(reset=True)       # the warm-up's exchanges (RCCL's lazy channel set-up among them) are not the timed ones
        par.collective_timing(True)            # HIP events on the library's stream around both exchanges (gloo dry run: wall clock)
    if not dry:
        torch.cuda.synchronize()
        pipe.ctx.prof_begin(args.prof_class)
        if rank == 0:
            sampler = CardSampler(dev.index).start()
    t_start = time.perf_counter()
    ms_denoise = ms_decode = 0.0
    res = None
    for _ in range(args.steps):
        res = one_batch()
        if res is not None:
            ms_denoise += res.msDenoise
            ms_decode += res.msDecode
    if not dry:
        torch.cuda.synchronize()
    par.barrier()
    elapsed = time.perf_counter() - t_start
    card = sampler.finish() if sampler is not None else None
    n_launch, prof_ms, prof_flops = (0, 0.0, 0.0) if dry else pipe.ctx.prof_end()
    elapsed = par.all_reduce_max(elapsed, dev)
    # per-exchange time on this rank's generating stream, max over ranks (a rank that arrives early waits inside the exchange)
    collective_ms = None
    if world > 1:
        st = par.collective_stats(reset=True)
        par.collective_timing(False)
        collective_ms = {}
        for name in ("broadcast", "all_gather"):
            per_batch = st[name]["ms"] / max(st[name]["calls"], 1)
            collective_ms[name] = round(par.all_reduce_max(per_batch, dev), 4)
            collective_ms[name + "_bytes"] = st[name]["bytes"] // max(st[name]["calls"], 1)
        collective_ms["measured"] = ("wall clock around the gloo collectives (dry run)" if dry else
                                     "HIP events on the context's stream around every exchange of the timed region") + \
                                    "; ms per batch, max over ranks (includes waiting for the slowest peer)"

    # ---- outside the timed region: the output is a real result (finite latents on every rank, a checksum of the batch)
    ok = 1.0
    if res is not None:
        ok = 1.0 if bool(torch.isfinite(res.latents).all()) else 0.0
    ok = -par.all_reduce_max(-ok, dev)                                               # min over ranks
    if rank != 0:
        par.finalize()
        return
    assert ok == 1.0, "non-finite latents after the timed region"
    batch_np = host_batch.numpy()
    checksum = f"{zlib.crc32(batch_np.tobytes()) & 0xffffffff:08x}"
    hist = None
    if not dry:
        import numpy as np
        hcount = np.bincount(batch_np.reshape(-1), minlength=256)
        hist = {"zero_frac": round(float(hcount[0]) / batch_np.size, 4), "full_frac": round(float(hcount[255]) / batch_np.size, 4),
                "levels_used": int((hcount > 0).sum())}
        assert hist["levels_used"] > 16, f"decoded image is degenerate: {hist}"

    K = args.steps
    # strength < 1 runs the tail of the schedule (E: 35 of 50): the generation reports it; the dry run asks the host scheduler
    if res is not None:
        T_eff = int(res.effectiveSteps)
    else:
        from flux2mi import api as _api
        T_eff = len(_api.set_timesteps(T, s_img, max(0.01, min(1.0, strength)))[0]) - 1
    if chain is not None:
        assert hook_calls[0] == (K + args.warmup) * T_eff, f"RePaint hook ran {hook_calls[0]} times for {(K + args.warmup) * T_eff} steps"
    total_denoise_steps = world * K * T_eff
    value = total_denoise_steps / elapsed
    sec_per_image = elapsed / K   # latency of one image (every rank produces one per batch)

    roofline = None
    step_ms = 0.0
    peak = 5000.0 if args.weights == "fp8" else 2500.0
    blocks = cfg["n_double"] + cfg["n_single"]
    S_all = S + (s_img if args.reference else 0)                # reference tokens ride along in every block (I2I, S = 8704)
    step_flops = blocks * (26.0 * S_all * d * d + 4.0 * S_all * S_all * d) + 2.0 * ((S_all - s_txt) * 128 * d + s_img * 128 * d + s_txt * cfg["joint_dim"] * d)
    if not dry:
        # ---- roofline of the dominant kernel: single-stream linear1 GEMM [S,d] x [9d,d]^T
        flops_per_launch = prof_flops / max(n_launch, 1)
        avg_ms = prof_ms / max(n_launch, 1)
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        # HBM-side traffic of that kernel from the committed PMC passes (rocprofv3 cannot run inside this
        # process); null when the summary is not there or another model / size / format is benchmarked
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "linear1_traffic.json")
        if args.model == "klein-4b" and (H, W) == (1024, 1024) and args.weights == "bf16" and os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f)["traffic_bytes_per_launch"]
        # what the matrix pipe of THIS card sustains at its power cap (register-only MFMA stream, ~0.6 s, after the timed region):
        # the ceiling a perfect kernel would see here; `peak` stays the data-sheet figure
        stream_tf = None
        if not args.no_mfma_stream:
            try:
                stream_tf = round(pipe.ctx.mfma_stream_tflops(fp8=(args.weights == "fp8")), 1)
            except Exception as e:   # diagnostic only
                print(f"[bench] mfma stream probe failed: {e}", file=sys.stderr)
        roofline = {"bound": "mfma", "kernel": pipe.ctx.prof_kernel_name(args.prof_class),
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "mfma_stream": stream_tf, "frac_of_mfma_stream": round(achieved / stream_tf, 4) if stream_tf else None,
                    "launches": n_launch, "avg_launch_ms": round(avg_ms, 4),
                    "flops_per_launch": flops_per_launch}
        step_ms = ms_denoise / (K * T_eff)

    # ---- the other kernels' rooflines: one extra SHORT generation per kernel class AFTER the timed region (same
    # model, same inputs, 4 denoise steps, HIP events around every launch of the class on the library's stream), and one decode for
    # the two VAE classes.  MFMA-bound classes against the dense peak, HBM-bound classes (work = algorithmic bytes) against 8 TB/s.
    roofline_secondary = roofline_hbm = None
    if not dry and not args.no_secondary:
        sec_traffic = None
        spath = os.path.join(ROOT, "profiles", "secondary_traffic.json")     # committed PMC passes (tools/secondary_traffic.py), Klein-4B 1024^2 bf16 only
        if args.model == "klein-4b" and (H, W) == (1024, 1024) and args.weights == "bf16" and not args.reference and os.path.exists(spath):
            with open(spath) as f:
                sec_traffic = {int(k): v["traffic_bytes_per_launch"] for k, v in json.load(f)["kernels"].items()}
        roofline_secondary, roofline_hbm = secondary_rooflines(pipe, torch, dev, H, W, peak, stream_tf, sec_traffic, lambda n: pipe.generate_with_result(
            "imageToImage" if ref_lat is not None else "textToImage", height=H, width=W, steps=n, guidance=guidance,
            precomputed_embeddings=emb_all[0], noise=noise, skip_decode=True, reference_latents=ref_lat, reference_ids=ref_ids))

    # ---- the decode tail with BOTH decoders (outside the timed region): the timed region uses --vae (default: standard, the
    # larger one); the reference pipeline's own default is the small decoder (Flux2Pipeline.swift:303)
    decode_by_vae = None
    if not dry:
        decode_by_vae = {v: time_vae_decode(pipe.ctx, v, syn, dev, H, W) for v in ("standard", "small")}
        if roofline_secondary is not None:
            # the whole decode against the MFMA peak on SURVEY §8(d)'s nominal FLOPs (1024^2: 10.475 / 5.997 TFLOP, scaled by the pixel count)
            for v, tf in (("standard", 10.475), ("small", 5.997)):
                fl = tf * (H * W) / (1024.0 * 1024.0)
                ach = fl / (decode_by_vae[v] * 1e-3)
                roofline_secondary.append({"kernel": f"VAE decode, whole ({v} decoder: every launch, latents -> u8 image)", "bound": "mfma",
                                           "achieved": round(ach, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4),
                                           "ms": decode_by_vae[v], "nominal_tflop": round(fl, 3),
                                           "measured": "synchronised wall time over 5 back-to-back decodes after the timed region"})

    cpu_baseline = None
    if world == 1 and not dry and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(P, cfg, syn, H, W, T, s_txt) if P is not None else \
            {"skipped": f"the oracle leg is timed for Klein-4B only ({args.model}: one fp32 CPU forward is minutes and the tensors do not fit the host)"}

    # ---- analytic scaling model (SURVEY §8e "Measurability"): 1 -> 8 GPUs from ONE image's time + the two exchanges + the batch D2H.
    # At N = 1 the base is the measured seconds/image; at N > 1 this run's own time minus its measured exchanges (what one replica
    # takes for its image), so that a measured curve and the model can be held against each other.  Dry run: a nominal 1 s base.
    base = 1.0 if dry else sec_per_image
    if world > 1 and collective_ms is not None and not dry:
        base = max(1e-6, sec_per_image - 1e-3 * (collective_ms["broadcast"] + collective_ms["all_gather"])
                   - (world - 1) * H * W * 3 / (par.PCIE_D2H_GBS * 1e9))
    scaling = par.scaling_model(base, T_eff, s_txt, J, H, W)
    scaling["base_from"] = ("nominal 1 s (dry run)" if dry else "measured seconds_per_image of this 1-GPU run" if world == 1 else
                            f"this {world}-GPU run's seconds_per_image minus its measured exchanges and the modelled extra D2H")

    out = {
        "metric": f"denoise-steps/sec (Flux.2 {args.model} {H}x{W} @ {T} steps; seconds/image reported alongside)"
                  + (" — DRY RUN: launcher/collectives self-test, no compute" if dry else ""),
        "value": None if dry else round(value, 4),
        "unit": "denoise-steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / K, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16" if args.weights == "bf16" else "fp8 (e4m3 block GEMMs with block-scaled activations, e4m3 attention Q K^T; bf16 P V, f32 residual)",
        "data": "synthetic",
        "config": {"workload": f"Flux.2 {args.model} {args.weights}, {H}x{W}, {T} denoise steps"
                               + (f" at strength {strength} ({T_eff} effective)" if strength < 1.0 else "")
                               + (" + RePaint step hook every step" if args.repaint else "")
                               + (f" + one {H}x{W} reference image (S = {S_all})" if args.reference else "")
                               + f" + {args.vae} VAE decode + image all-gather + D2H, 1 image per GPU per bench step",
                   "baseline_config": f"{args.config} = BASELINE.json {CONFIGS[args.config]['name']}",
                   "matches_baseline_config": all(getattr(args, k) == v for k, v in CONFIGS[args.config].items() if k != "name"),
                   "images_per_bench_step": world, "s_txt": s_txt, "s_img": s_img, "effective_denoise_steps": T_eff,
                   "parallelism": f"dp{world}"},
        "seconds_per_image": round(sec_per_image, 4),
        "denoise_ms_per_denoise_step": round(step_ms, 3),
        "decode_ms_per_image": round(ms_decode / K, 3),
        "decode_vae": args.vae,
        "decode_ms_by_vae": decode_by_vae,
        "seconds_per_image_by_vae": None if not decode_by_vae else
            {v: round(sec_per_image + (ms - ms_decode / K) * 1e-3, 4) for v, ms in decode_by_vae.items()},
        "dit_step_tflops": round(step_flops / (step_ms * 1e-3) / 1e12, 2) if step_ms > 0 else None,
        "dit_step_mfma_frac": round(step_flops / (step_ms * 1e-3) / 1e12 / peak, 4) if step_ms > 0 else None,
        "attention_softmax": None if dry else pipe.transformer.attention_softmax_paths(),
        # KV pieces per (head, query block) item of the attention's partial last round on this GPU (1 = whole items only)
        "attention_kv_split_pieces": None if dry else pipe.ctx.op_attention_split_info(S_all, cfg["heads"])[0],
        "output_check": {"latents_finite": True, "image_crc32": checksum, "u8_histogram": hist,
                         "weights": "synthetic.LazyDitWeights(seed 1234) + make_vae_weights(seed 4321)"},
        "model_load_s": round(t_load, 2),
        "collective_ms": collective_ms,
        "scaling_model": scaling,
        "roofline": roofline,
        "roofline_secondary": roofline_secondary,
        "roofline_hbm": roofline_hbm,
        "card": card,
        "cpu_baseline": cpu_baseline,
    }
    if dry:
        out["dry_run"] = True
    print(json.dumps(out), flush=True)
    par.finalize()


def secondary_rooflines(pipe, torch, dev, H, W, peak, stream_tf, traffic, short_generation, steps=4):
    """[MFMA-bound entries], [HBM-bound entries]: achieved = work of the class's launches / their summed HIP-event time."""
    ctx = pipe.ctx
    where = f"one extra {steps}-step generation after the timed region (same model and inputs; HIP events around every launch of the class)"

    def entry(cls, n, ms, work, unit, pk, extra=None):
        ach = work / max(ms * 1e-3, 1e-12) / (1e12 if unit == "TFLOP/s" else 1e9)
        e = {"kernel": ctx.prof_kernel_name(cls), "bound": "mfma" if unit == "TFLOP/s" else "hbm", "achieved": round(ach, 2), "peak": pk,
             "unit": unit, "frac": round(ach / pk, 4), "launches": n, "avg_launch_ms": round(ms / max(n, 1), 4),
             "work_per_launch": work / max(n, 1), "traffic": (traffic or {}).get(cls), "measured": where}
        if unit == "TFLOP/s" and stream_tf:
            e["frac_of_mfma_stream"] = round(ach / stream_tf, 4)
        e.update(extra or {})
        return e

    mfma, hbm = [], []
    for cls in (2, 4, 7, 5):                          # attention, linear2, proj + mlp_out, LayerNorm + modulate
        ctx.prof_begin(cls)
        short_generation(steps)
        n, ms, work = ctx.prof_end()
        if n:
            (hbm if cls == 5 else mfma).append(entry(cls, n, ms, work, "GB/s" if cls == 5 else "TFLOP/s", 8000.0 if cls == 5 else peak))
    if pipe.vae is not None:
        g = torch.Generator(device=dev)
        g.manual_seed(11)
        z = torch.randn((1, 32, H // 8, W // 8), generator=g, device=dev, dtype=torch.float32)
        pipe.vae.decode(z, want_f32=False)
        for cls in (3, 6, 8):                         # VAE convolutions, GroupNorm apply, the mid block's streaming attention
            ctx.prof_begin(cls)
            pipe.vae.decode(z, want_f32=False)
            n, ms, work = ctx.prof_end()
            if n:
                one = {"measured": "one extra decode after the timed region (HIP events around every launch of the class)"}
                (hbm if cls == 6 else mfma).append(entry(cls, n, ms, work, "GB/s" if cls == 6 else "TFLOP/s", 8000.0 if cls == 6 else 2500.0, one))
    return mfma, hbm


class CardSampler:
    """Read-only samples of the card's power and shader clock (amdgpu hwmon sysfs of the PCI device torch calls cuda:<index>) every
    0.1 s during the timed region, on rank 0: the hot kernels run at the board's power cap, so the clock they get — and with it
    `roofline.achieved` — differs from box to box.  Everything is optional: missing files give nulls."""

    def __init__(self, dev_index):
        import glob
        import threading
        self.hw = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(dev_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            hits = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            self.hw = hits[0] if hits else None
        except Exception:
            self.hw = None
        self.stop = threading.Event()
        self.rows = []
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _read(self, name):
        try:
            with open(os.path.join(self.hw, name)) as f:
                return float(f.read().strip())
        except (OSError, ValueError, TypeError):
            return None

    def _run(self):
        while not self.stop.is_set():
            pw = self._read("power1_average")
            if pw is None:
                pw = self._read("power1_input")
            self.rows.append((pw, self._read("freq1_input")))
            self.stop.wait(0.1)

    def start(self):
        if self.hw:
            self.thread.start()
        return self

    def finish(self):
        if not self.hw:
            return None
        self.stop.set()
        self.thread.join()
        pw = sorted(r[0] / 1e6 for r in self.rows if r[0])
        ck = sorted(r[1] / 1e6 for r in self.rows if r[1])
        cap = self._read("power1_cap")
        med = lambda v: round(v[len(v) // 2], 1) if v else None
        return {"cap_w": round(cap / 1e6, 1) if cap else None, "median_w": med(pw), "max_w": round(pw[-1], 1) if pw else None,
                "median_sclk_mhz": med(ck), "min_sclk_mhz": round(ck[0], 1) if ck else None, "samples": len(self.rows)}


def time_vae_decode(ctx, variant, syn, dev, H, W, reps=5):
    """ms per latents -> u8 image decode (AutoencoderKLFlux2.decode + post-processing) of one VAE variant, HIP path, synchronised wall
    time over `reps` back-to-back decodes after one warm-up."""
    import torch

    from flux2mi.api import VAE
    vcfg = syn.VAE_PRESETS[variant]
    vae = VAE(ctx, vcfg).load(syn.make_vae_weights(vcfg, seed=4321, device=dev, encoder=False)).finalize()
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    z = torch.randn((1, 32, H // 8, W // 8), generator=g, device=dev, dtype=torch.float32)
    vae.decode(z, want_f32=False)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        vae.decode(z, want_f32=False)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    vae.close()
    return round(dt * 1e3, 3)


def granted_cores() -> int:
    """Host cores this process may use: the cgroup CPU quota where there is one, else the visible CPUs."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def run_cpu_baseline(P_dev, cfg, syn, H, W, T, s_txt):
    """The restated oracle (torch CPU fp32 — NOT the reference's MLX path, which cannot run here) on a bounded
    sample of THE metric's workload: ONE denoise step (one DiT forward at S = s_txt + HW/256) of the same model,
    size and inputs; the image time is that x T (extrapolated, no VAE).  BASELINE.md §3.  A 256x256 forward
    (config A's step) is timed alongside."""
    import torch

    from oracle import flux2_ref as R

    threads = host_threads()
    cores = granted_cores()
    torch.set_num_threads(threads)
    Pc = {k: v.cpu() for k, v in P_dev.items()}

    def one(h, w):
        inp = syn.make_inputs(cfg, h, w, s_txt=s_txt)
        lat = R.pack_patchified_to_sequence(inp["noise"])
        ii, ti = R.image_position_ids(h, w), R.text_position_ids(s_txt)
        t0 = time.perf_counter()
        with torch.no_grad():
            R.dit_forward(Pc, cfg, lat, inp["txt"].float(), torch.tensor([0.9]), torch.tensor([4.0]) if cfg["guidance"] else None, ii, ti)
        return time.perf_counter() - t0

    dt_small = one(256, 256)
    dt = one(H, W)
    return {"value": round(1.0 / dt, 5), "unit": "denoise-steps/s", "cores": cores, "cores_granted": cores, "threads": threads, "kind": "port",
            "sample": f"1 denoise step (DiT forward) of the restated fp32 oracle (torch CPU, {threads} threads on the {cores} host cores "
                      f"the cgroup quota grants), same model / {H}x{W} / S={s_txt + (H // 16) * (W // 16)} as the metric",
            "seconds_per_step": round(dt, 2), "extrapolated": True, "includes_vae": False,
            "seconds_per_image_extrapolated": round(dt * T, 1),
            "sample_256": {"seconds_per_step": round(dt_small, 2), "steps_per_s": round(1.0 / dt_small, 4)}}


if __name__ == "__main__":
    main()
