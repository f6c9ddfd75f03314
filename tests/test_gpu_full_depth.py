"""-m gpu: BASELINE.json configurations C, D and E at their FULL DEPTH.

  Dev-32B      8 double + 48 single blocks, d 6144, 48 heads, J 15360, guidance embedder  (Flux2Config.swift:258-291)
  Klein-9B     8 double + 24 single blocks, d 4096, 32 heads, J 12288, e4m3 block GEMMs    (Flux2Config.swift:295-329)

What a one-plus-one-block slice at those widths (tests/test_gpu_configs.py) never touches: 64 GB / 18 GB of resident weights and
their allocation sizes, byte offsets beyond 2^31 in the fused weight buffers and in `comb` (S 8704 x 4 d), 56 blocks of f32
residual accumulation, the workspaces of S 8704 at d 6144.

Each model is built ONCE per module from per-tensor-seeded synthetic weights (`synthetic.LazyDitWeights`: drawn on the device one
tensor at a time — the loader and the CPU oracle see bit-identical bf16 values without 64 GB ever being resident on the host) and is
checked two ways:
  * one forward at 256 x 256 (S = 512 + 256) against the fp32 CPU oracle (Dev: ~42 TFLOP of f32 on the host cores), Klein-9B against
    the oracle running the same e4m3 emulation;
  * the whole configuration at its real size — C: 1024^2, 50 steps; D: 1024^2, 28 steps, fp8; E: strength 0.7 of 50 = 35 steps with
    the RePaint hook, and one step at S 8704 with a 1024^2 reference — through size-independent properties: finite, bit-reproducible
    latents and image bytes, the effective step counts, the hook calls and their sigmas, the source restored exactly outside the mask.
Match: Flux2Pipeline.swift:1696-1808 (I2I loop), :1933-2052 (T2I loop), Flux2MaskedInpaintingChain.swift:399-431."""
import numpy as np
import pytest
import torch

from oracle import flux2_ref as R
from tests.util import oracle_cached, rel_l2, tapped_stream

pytestmark = pytest.mark.gpu


def _pipe(flux2mi, preset, quantization, seed):
    from flux2mi import synthetic as syn
    from flux2mi.pipeline import Flux2Pipeline
    cfg, vcfg = syn.DIT_PRESETS[preset], syn.VAE_PRESETS["standard"]
    W = syn.LazyDitWeights(cfg, seed=seed, device="cuda")
    VP = syn.make_vae_weights(vcfg, encoder=False)
    pipe = Flux2Pipeline(model=preset, vae_variant="standard", quantization=quantization).load_models(W, VP)
    return dict(pipe=pipe, W=W, cfg=cfg, VP=VP, vcfg=vcfg)


@pytest.fixture(scope="module")
def dev32b(flux2mi, ctx):
    k = _pipe(flux2mi, "dev", "bf16", 61)
    yield k
    k["pipe"].clearAll()


@pytest.fixture(scope="module")
def klein9b_fp8(flux2mi, ctx):
    k = _pipe(flux2mi, "klein-9b", "fp8", 62)
    yield k
    k["pipe"].clearAll()


def _forward_256(k, t, guidance):
    from flux2mi import synthetic as syn
    cfg = k["cfg"]
    inp = syn.make_inputs(cfg, 256, 256, s_txt=512)
    lat = R.pack_patchified_to_sequence(inp["noise"])
    ii, ti = R.image_position_ids(256, 256), R.text_position_ids(512)
    out = k["pipe"].transformer.forward(lat.cuda(), inp["txt"].cuda(), t, guidance, ii.cuda(), ti.cuda()).cpu()
    g = torch.tensor([guidance]) if cfg["guidance"] else None
    return out, (k["W"].cpu(), cfg, lat, inp["txt"].float(), torch.tensor([t]), g, ii, ti)


# ------------------------------------------------------------------------------------------------ C: Dev-32B
def test_dev32b_full_depth_forward_vs_oracle(dev32b):
    """Dev at full depth (8 + 48 blocks), one forward at 256 x 256 (S 768) against the fp32 oracle: 3e-2 like the Klein-4B full-depth
    forward (measured value printed)."""
    import time
    out, args = _forward_256(dev32b, 0.71, 4.0)
    t0 = time.time()

    def oracle():
        with torch.no_grad():
            return {"ref": R.dit_forward(*args)[0]}
    # (seeded case: the 60 s oracle forward is a committed fixture, used while the inputs' fingerprint matches — tests/util.py)
    W = dev32b["W"]
    got, cached = oracle_cached("dev32b_forward_256", [W["img_in.weight"], W["single_blocks.47.linear2.weight"][:16], W["double_blocks.7.img_mlp.0.weight"][:16],
                                                       W["guidance_in.in_layer.weight"],
                                                       args[2], args[3]], oracle)
    e = rel_l2(out, got["ref"])
    print(f"Dev-32B full depth, S 768: rel-L2 {e:.3e} (oracle {'from the committed fixture' if cached else f'run live, {time.time() - t0:.0f} s'})")
    assert torch.isfinite(out).all()
    assert e < 3e-2, e


def test_config_c_dev32b_1024_50_steps_properties(dev32b):
    """BASELINE config C whole: Dev-32B bf16, 1024 x 1024, 50 steps, guidance 4 + standard VAE decode."""
    from flux2mi import synthetic as syn
    k = dev32b
    inp = syn.make_inputs(k["cfg"], 1024, 1024, s_txt=512)
    seen, runs = [], []
    for r in range(2):
        runs.append(k["pipe"].generate_with_result("textToImage", height=1024, width=1024, steps=50, guidance=4.0,
                                                   precomputed_embeddings=inp["txt"], noise=inp["noise"], return_latents=True,
                                                   on_step=(lambda c, x: seen.append((c.stepIdx, c.totalSteps, c.sigma, c.sigmaNext)) or x) if r == 0 else None))
    a, b = runs
    assert a.effectiveSteps == 50 and len(seen) == 50 and seen[-1][0] == 49 and seen[-1][3] == 0.0 and seen[0][2] == 1.0
    assert all(s[2] > s[3] for s in seen)
    assert torch.isfinite(a.latents).all()
    assert torch.equal(a.latents, b.latents) and np.array_equal(a.image, b.image)
    assert a.image.shape == (1024, 1024, 3) and a.image.dtype == np.uint8
    hist = np.bincount(a.image.reshape(-1), minlength=256) / a.image.size
    assert hist[0] < 0.5 and hist[255] < 0.5 and (hist > 0).sum() > 64
    rms = float(a.latents.float().pow(2).mean().sqrt())
    print(f"config C: {a.msDenoise / 50:.1f} ms per step, decode {a.msDecode:.1f} ms, latent RMS {rms:.3f}")
    assert 0.05 < rms < 20.0, rms
    # guidance is live at full depth: another guidance value gives another image
    c = k["pipe"].generate_with_result("textToImage", height=1024, width=1024, steps=2, guidance=1.0, precomputed_embeddings=inp["txt"],
                                       noise=inp["noise"], return_latents=True, skip_decode=True)
    d = k["pipe"].generate_with_result("textToImage", height=1024, width=1024, steps=2, guidance=4.0, precomputed_embeddings=inp["txt"],
                                       noise=inp["noise"], return_latents=True, skip_decode=True)
    assert rel_l2(c.latents, d.latents) > 1e-4


def test_dev32b_bf16_teacher_forced_blocks_vs_store_point_oracle(dev32b):
    """Dev-32B block by block (8 double + 48 single, d 6144, 48 heads, guidance embedder), like Klein-4B / Klein-9B: every block is fed the
    input the oracle's store-point emulation (`device_roundings`) gave it, through flux2mi_dit_forward_block, and must reproduce the
    emulation's increment and output.  S = 64 text + 256 image tokens (the 56 tapped streams of a 768-token forward are 1 GB and its
    emulating oracle two minutes of host time: the block arithmetic does not depend on S); the oracle runs live (~45 s).
    Measured (profiles/r06_f_gpu_tests.log): single blocks like Klein's — increment 2.8e-4 ... 3.5e-4, output <= 9.6e-5 (bounds 5e-4 /
    3e-4); the double blocks of this width (d 6144, 18432-wide MLP, six bf16 store points in sequence) sit at 1.3e-3 ... 2.4e-3 on
    the increment and 4.5e-4 ... 1.5e-3 on the output (the first blocks' streams are small against their increments), 1.5x Klein-9B's:
    bounds 4e-3 / 2.5e-3.  A scale applied twice, swapped halves or a stale buffer is an O(1) increment error."""
    import time
    from flux2mi import synthetic as syn
    k = dev32b
    cfg = k["cfg"]
    s_txt = 64
    inp = syn.make_inputs(cfg, 256, 256, s_txt=s_txt)
    lat = R.pack_patchified_to_sequence(inp["noise"])
    ii, ti = R.image_position_ids(256, 256), R.text_position_ids(s_txt)
    Pc = k["W"].cpu()
    taps = {}
    t0 = time.time()
    with torch.no_grad(), R.device_roundings():
        R.dit_forward(Pc, cfg, lat, inp["txt"].float(), torch.tensor([0.6]), torch.tensor([4.0]), ii, ti, taps=taps)
    t_oracle = time.time() - t0
    nd, ns = cfg["n_double"], cfg["n_single"]
    dit = k["pipe"].transformer
    rows = []
    for layer in range(nd + ns):
        x_in, y_ref = tapped_stream(taps, layer - 1, nd), tapped_stream(taps, layer, nd)
        y = dit.forward_block(layer, x_in.cuda(), inp["txt"].cuda(), 0.6, 4.0, ii.cuda(), ti.cuda()).cpu()
        rows.append((layer, rel_l2(y, y_ref), rel_l2(y - x_in, y_ref - x_in)))
    print(f"Dev-32B bf16 teacher-forced blocks, S {s_txt + 256} (oracle {t_oracle:.0f} s; layer: output / increment): "
          + ", ".join(f"{l}: {a:.1e} / {b:.1e}" for l, a, b in rows))
    assert max(r[1] for r in rows[:nd]) < 2.5e-3 and max(r[2] for r in rows[:nd]) < 4e-3, rows[:nd]
    assert max(r[1] for r in rows[nd:]) < 3e-4 and max(r[2] for r in rows[nd:]) < 5e-4, rows[nd:]


# ------------------------------------------------------------------------------------------------ E: Dev-32B, strength 0.7, RePaint
def _repaint_inputs(S):
    g = torch.Generator().manual_seed(99)
    image_latents = torch.randn((1, S, 128), generator=g)
    blend_noise = torch.randn((1, S, 128), generator=g)
    side = int(S ** 0.5)
    m = torch.zeros((side, side))
    m[side // 4:3 * side // 4, side // 4:3 * side // 4] = 1.0          # centred square: 25 % of the tokens
    m[side // 4 - 1, side // 4:3 * side // 4] = 0.5                       # soft 1-token ramp
    return image_latents, blend_noise, m.reshape(1, S, 1)


def test_config_e_dev32b_strength_repaint_1024_properties(dev32b):
    """BASELINE config E whole, T2I + initLatents form (S 4608): Dev-32B, 1024 x 1024, strength 0.7 of 50 steps = 35 effective steps
    (FlowMatchEulerScheduler.swift:92-103), the RePaint blend on every step (Flux2MaskedInpaintingChain.swift:399-403)."""
    from flux2mi import synthetic as syn
    from flux2mi.pipeline import Flux2MaskedInpaintingChain
    k = dev32b
    inp = syn.make_inputs(k["cfg"], 1024, 1024, s_txt=512)
    image_latents, blend_noise, mask = _repaint_inputs(4096)
    chain = Flux2MaskedInpaintingChain(k["pipe"], image_latents, mask, blend_noise)
    runs = [chain.run(inp["txt"], 1024, 1024, steps=50, guidance=4.0, strength=0.7, noise=inp["noise"], return_latents=True) for _ in range(2)]
    a, b = runs
    assert a.effectiveSteps == 35
    lat = a.latents.cpu()
    assert torch.isfinite(lat).all()
    assert torch.equal(a.latents, b.latents) and np.array_equal(a.image, b.image)
    keep = (mask == 0).expand_as(lat)
    assert torch.equal(lat[keep], image_latents[keep])                   # sigma_next = 0 on the last call: the source restored exactly
    inside = (mask == 1).expand_as(lat)
    assert not torch.equal(lat[inside], image_latents[inside])
    # the hook itself: 35 calls, sigma_next = 0 on the last, every call sees the output tokens only
    seen = []
    k["pipe"].generate_with_result("textToImage", height=1024, width=1024, steps=50, guidance=4.0, precomputed_embeddings=inp["txt"],
                                   noise=inp["noise"], init_latents=image_latents, strength=0.7, skip_decode=True,
                                   on_step=lambda c, x: seen.append((c.stepIdx, c.totalSteps, c.sigmaNext, tuple(x.shape), c.isI2I)) or x)
    assert len(seen) == 35 and seen[-1][2] == 0.0 and seen[0][1] == 35 and all(s[3] == (1, 4096, 128) and not s[4] for s in seen)
    print(f"config E (S 4608): {a.msDenoise / 35:.1f} ms per step")


def test_config_e_dev32b_reference_sequence_8704(dev32b):
    """Config E with one 1024 x 1024 reference image on the I2I path AT DEV DEPTH: S = 512 + 4096 + 4096 = 8704 in all 56 blocks
    (`comb` = 8704 x 24576 bf16 = 428 MB, linear1 output 8704 x 55296), strength 0.7 of 5 steps + the RePaint hook.  Properties: finite,
    bit-reproducible, the hook sees [1, 4096, 128] with isI2I, the source restored outside the mask; and the reference tokens matter
    (another reference image gives other latents inside the mask)."""
    from flux2mi import synthetic as syn
    from flux2mi.pipeline import Flux2MaskedInpaintingChain
    k = dev32b
    inp = syn.make_inputs(k["cfg"], 1024, 1024, s_txt=512)
    image_latents, blend_noise, mask = _repaint_inputs(4096)
    g = torch.Generator().manual_seed(5)
    ref_lat = torch.randn((1, 4096, 128), generator=g)
    ref_ids = R.reference_position_ids([64], [64])
    chain = Flux2MaskedInpaintingChain(k["pipe"], image_latents, mask, blend_noise)
    run = lambda rl: chain.run(inp["txt"], 1024, 1024, steps=5, guidance=4.0, strength=0.7, noise=inp["noise"], reference_latents=rl,
                               reference_ids=ref_ids, return_latents=True)
    a, b, c = run(ref_lat), run(ref_lat), run(ref_lat.flip(1))
    assert a.effectiveSteps >= 3
    lat = a.latents.cpu()
    assert torch.isfinite(lat).all() and torch.equal(a.latents, b.latents) and np.array_equal(a.image, b.image)
    keep = (mask == 0).expand_as(lat)
    assert torch.equal(lat[keep], image_latents[keep])
    assert rel_l2(c.latents, a.latents) > 1e-4
    seen = []
    k["pipe"].generate_with_result("imageToImage", height=1024, width=1024, steps=5, guidance=4.0, precomputed_embeddings=inp["txt"],
                                   noise=inp["noise"], init_latents=image_latents, strength=0.7, skip_decode=True, reference_latents=ref_lat,
                                   reference_ids=ref_ids, on_step=lambda c_, x: seen.append((tuple(x.shape), c_.isI2I, c_.sigmaNext)) or x)
    assert seen and all(s[0] == (1, 4096, 128) and s[1] for s in seen) and seen[-1][2] == 0.0
    print(f"config E (S 8704): {a.msDenoise / a.effectiveSteps:.1f} ms per step")


# ------------------------------------------------------------------------------------------------ D: Klein-9B fp8
def _oracle_block(P, cfg, layer, x, taps, s_txt):
    """Block `layer` of the oracle (in whatever emulation mode is active) on the residual stream x [S, d] (text rows first)."""
    nd, H = cfg["n_double"], cfg["heads"]
    temb, cos, sin = taps["temb"], taps["cos"], taps["sin"]
    if layer < nd:
        img_mod = R.modulation(temb, P["double_stream_modulation_img.lin.weight"], 2)
        txt_mod = R.modulation(temb, P["double_stream_modulation_txt.lin.weight"], 2)
        txt, img = R.double_block(P, layer, x[s_txt:].unsqueeze(0), x[:s_txt].unsqueeze(0), H, cos, sin, img_mod, txt_mod)
        return torch.cat([txt[0], img[0]], 0)
    smod = R.modulation(temb, P["single_stream_modulation.lin.weight"], 1)
    return R.single_block(P, layer - nd, x.unsqueeze(0), H, cos, sin, smod)[0]


def test_klein9b_full_depth_forward_bf16_and_fp8_vs_oracles(klein9b_fp8, flux2mi, ctx):
    """Klein-9B at full depth (8 + 24 blocks), one forward at 256 x 256.  bf16 weights against the fp32 oracle (3e-2, like the other
    full-depth forwards): the clean check of the 32-block structure.  e4m3 block GEMMs against the oracle running the same e4m3
    emulation with the HIP path's bf16 store points (`fp8_block_gemms(device_roundings=True)`): with these synthetic N(0, 0.02) weights
    32 blocks amplify the quantisation noise to q ~ 0.26 of the output, and roundings cascade (a 1e-7 perturbation of a block's input
    moves its output by 2e-4 whatever its size: profiles/r04_d_store_point_probe.txt), so over 32 blocks two correct implementations
    de-correlate — the free-running statement stays relative to q (< 0.9 q, effect reproduced in size and direction).

    The SHARP statement is TEACHER-FORCED: every one of the 32 blocks is fed the emulation's own block input through
    flux2mi_dit_forward_block and must reproduce the emulation's block output to <= 3e-3 and its INCREMENT (output - input: what the
    block computes) to within 0.3 of that block's own quantisation effect q_b = 6.6e-2 (the emulation's increment against the f32
    oracle's on the same input) — single blocks, one quantisation stage deep, to within 0.06 q_b.  Measured
    (profiles/r04_e_parity_tests.log): double blocks 1.1e-2 ... 1.4e-2 = 0.17 ... 0.21 q_b (the MLP half re-quantises what the
    attention half changed: two stages in sequence), single blocks 1.3e-3 ... 2.2e-3 = 0.02 ... 0.034 q_b; outputs <= 1.6e-3.
    Anything wrong between the ops of a block — a scale applied twice, halves swapped, a stale buffer — is an O(1) increment error; it
    cannot hide behind 32 blocks of accumulated noise."""
    from flux2mi import synthetic as syn
    from flux2mi.api import DiT
    k = klein9b_fp8
    out8, args = _forward_256(k, 0.6, 0.0)
    dit16 = DiT(ctx, "klein-9b").load(k["W"]).finalize()
    Pc, cfg, lat, _, _, _, ii, ti = args
    inp = syn.make_inputs(cfg, 256, 256, s_txt=512)
    out16 = dit16.forward(lat.cuda(), inp["txt"].cuda(), 0.6, 0.0, ii.cuda(), ti.cuda()).cpu()
    dit16.close()
    taps8 = {}
    with torch.no_grad():
        ref32 = R.dit_forward(*args)[0]
        with R.fp8_block_gemms(device_roundings=True):
            ref8 = R.dit_forward(*args, taps=taps8)[0]
    e16 = rel_l2(out16, ref32)
    q, e8, e32 = rel_l2(ref8, ref32), rel_l2(out8, ref8), rel_l2(out8, ref32)
    d_hip, d_ref = (out8 - out16).flatten(), (ref8 - ref32).flatten()
    cos = float(torch.dot(d_hip, d_ref) / (d_hip.norm() * d_ref.norm()))
    print(f"Klein-9B full depth, S 768: bf16 vs f32 oracle {e16:.3e}; fp8 vs emulating oracle {e8:.3e} = {e8 / q:.2f} q, vs f32 oracle {e32:.3e}, "
          f"quantisation effect q {q:.3e}, direction cosine {cos:.3f}")
    assert torch.isfinite(out8).all() and torch.isfinite(out16).all()
    assert e16 < 3e-2, e16
    assert e8 < 0.9 * q, (e8, q)
    assert 0.5 * q < e32 < 1.5 * q, (e32, q)
    assert cos > 0.5, cos

    # ---- teacher-forced, block by block: the emulation's block input in, the HIP block's output against the emulation's
    nd, ns = cfg["n_double"], cfg["n_single"]
    dit8 = k["pipe"].transformer
    rows = []
    for layer in range(nd + ns):
        x_in, y_ref = tapped_stream(taps8, layer - 1, nd), tapped_stream(taps8, layer, nd)
        y = dit8.forward_block(layer, x_in.cuda(), inp["txt"].cuda(), 0.6, 0.0, ii.cuda(), ti.cuda()).cpu()
        with torch.no_grad():
            y32 = _oracle_block(Pc, cfg, layer, x_in, taps8, 512)                     # the un-quantised block on the same input
        q_b = rel_l2(y_ref - x_in, y32 - x_in)
        rows.append((layer, rel_l2(y, y_ref), rel_l2(y - x_in, y_ref - x_in), q_b))
    print("teacher-forced blocks (layer: rel-L2 of the output / of the increment / the block's own quantisation effect q_b): "
          + ", ".join(f"{l}: {a:.1e} / {b:.1e} / {c:.1e}" for l, a, b, c in rows))
    assert max(r[1] for r in rows) < 3e-3, rows
    assert all(r[2] < 0.3 * r[3] for r in rows), rows
    assert all(r[2] < 0.06 * r[3] for r in rows[nd:]), rows[nd:]


def test_klein9b_bf16_teacher_forced_blocks_vs_store_point_oracle(klein9b_fp8, flux2mi, ctx):
    """The bf16 model, every block teacher-forced against the oracle with the HIP path's bf16 store points (`device_roundings`): the
    increment of each of the 32 blocks within 2e-3 of the emulation's (double blocks, six bf16 store points in sequence: measured
    9.5e-4 ... 1.1e-3) and within 5e-4 (single blocks, three: 1.8e-4 ... 2.4e-4), its output within 3e-4 (<= 1.3e-4).  Against the
    plain f32 oracle the same increments sit ~4e-3 away (the bf16 storage itself), which is why a sharp bf16 check needs the
    store-point emulation (profiles/r04_e_parity_tests.log)."""
    from flux2mi import synthetic as syn
    from flux2mi.api import DiT
    k = klein9b_fp8
    cfg = k["cfg"]
    inp = syn.make_inputs(cfg, 256, 256, s_txt=512)
    lat = R.pack_patchified_to_sequence(inp["noise"])
    ii, ti = R.image_position_ids(256, 256), R.text_position_ids(512)
    Pc = k["W"].cpu()
    taps = {}
    with torch.no_grad(), R.device_roundings():
        R.dit_forward(Pc, cfg, lat, inp["txt"].float(), torch.tensor([0.6]), None, ii, ti, taps=taps)
    dit16 = DiT(ctx, "klein-9b").load(k["W"]).finalize()
    nd, ns = cfg["n_double"], cfg["n_single"]
    rows = []
    for layer in range(nd + ns):
        x_in, y_ref = tapped_stream(taps, layer - 1, nd), tapped_stream(taps, layer, nd)
        y = dit16.forward_block(layer, x_in.cuda(), inp["txt"].cuda(), 0.6, 0.0, ii.cuda(), ti.cuda()).cpu()
        rows.append((layer, rel_l2(y, y_ref), rel_l2(y - x_in, y_ref - x_in)))
    dit16.close()
    print("bf16 teacher-forced blocks (layer: output / increment): " + ", ".join(f"{l}: {a:.1e} / {b:.1e}" for l, a, b in rows))
    assert max(r[1] for r in rows) < 3e-4, rows
    assert max(r[2] for r in rows) < 2e-3, rows
    assert max(r[2] for r in rows[nd:]) < 5e-4, rows[nd:]


def test_config_d_klein9b_fp8_1024_28_steps_properties(klein9b_fp8):
    """BASELINE config D per GPU: Klein-9B fp8, 1024 x 1024, 28 steps + standard VAE decode (the batch of 8 is 8 such replicas)."""
    from flux2mi import synthetic as syn
    k = klein9b_fp8
    inp = syn.make_inputs(k["cfg"], 1024, 1024, s_txt=512)
    seen, runs = [], []
    for r in range(2):
        runs.append(k["pipe"].generate_with_result("textToImage", height=1024, width=1024, steps=28, precomputed_embeddings=inp["txt"],
                                                   noise=inp["noise"], return_latents=True,
                                                   on_step=(lambda c, x: seen.append((c.stepIdx, c.sigmaNext)) or x) if r == 0 else None))
    a, b = runs
    assert a.effectiveSteps == 28 and len(seen) == 28 and seen[-1] == (27, 0.0)
    assert torch.isfinite(a.latents).all()
    assert torch.equal(a.latents, b.latents) and np.array_equal(a.image, b.image)
    hist = np.bincount(a.image.reshape(-1), minlength=256) / a.image.size
    assert hist[0] < 0.5 and hist[255] < 0.5 and (hist > 0).sum() > 64
    print(f"config D: {a.msDenoise / 28:.1f} ms per step, decode {a.msDecode:.1f} ms")
