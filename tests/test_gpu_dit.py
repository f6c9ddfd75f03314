"""-m gpu: DiT forward parity, HIP path (C ABI) vs the fp32 oracle on identical
bf16-rounded synthetic weights and inputs.

Tolerance (stated): rel-L2 of the [S_img,128] prediction <= 2e-2 vs the fp32 oracle
(bf16 GEMM operands / attention probabilities, f32 accumulation and residual stream)."""
import pytest
import torch

from oracle import flux2_ref as R
from tests.util import rel_l2, to_dev

pytestmark = pytest.mark.gpu


def _forward_both(flux2mi, ctx, cfg, s_txt, h, w, t=0.731, g=4.0, seed=1234, refs=None):
    from flux2mi import synthetic as syn
    from flux2mi.api import DiT

    P = syn.make_dit_weights(cfg, seed=seed)
    inp = syn.make_inputs(cfg, h, w, s_txt=s_txt)
    lat = R.pack_patchified_to_sequence(inp["noise"])
    img_ids = R.image_position_ids(h, w)
    if refs is not None:
        lat = torch.cat([lat, refs[0]], dim=1)
        img_ids = torch.cat([img_ids, refs[1]], dim=0)
    txt_ids = R.text_position_ids(s_txt)
    dit = DiT(ctx, cfg).load(P).finalize()
    out = dit.forward(lat.cuda(), inp["txt"].cuda(), t, g, img_ids.cuda(), txt_ids.cuda())
    gt = torch.tensor([g]) if cfg["guidance"] else None
    ref = R.dit_forward(P, cfg, lat, inp["txt"].float(), torch.tensor([t]), gt, img_ids, txt_ids)[0]
    dit.close()
    return out, ref


def test_tiny_forward(flux2mi, ctx):
    from flux2mi import synthetic as syn
    out, ref = _forward_both(flux2mi, ctx, syn.DIT_PRESETS["tiny"], s_txt=16, h=128, w=128)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 2e-2


def test_tiny_forward_ragged_sequence(flux2mi, ctx):
    """S = 19 + 35 = 54: nothing is a multiple of a tile."""
    from flux2mi import synthetic as syn
    out, ref = _forward_both(flux2mi, ctx, syn.DIT_PRESETS["tiny"], s_txt=19, h=80, w=112)
    assert rel_l2(out, ref) < 2e-2


def test_tiny_forward_guidance_embeds(flux2mi, ctx):
    from flux2mi import synthetic as syn
    out, ref = _forward_both(flux2mi, ctx, syn.DIT_PRESETS["tiny-g"], s_txt=16, h=128, w=128, g=4.0)
    assert rel_l2(out, ref) < 2e-2


def test_tiny_forward_reference_tokens(flux2mi, ctx):
    """I2I conditioning: reference tokens appended after the output tokens with T = 10."""
    from flux2mi import synthetic as syn
    g = torch.Generator().manual_seed(5)
    ref_lat = torch.randn((1, 24, 128), generator=g)
    ref_ids = R.reference_position_ids([4], [6])
    out, ref = _forward_both(flux2mi, ctx, syn.DIT_PRESETS["tiny"], s_txt=16, h=128, w=128, refs=(ref_lat, ref_ids))
    assert out.shape[0] == 64 + 24
    assert rel_l2(out, ref) < 2e-2


def test_diffusers_key_scheme_equals_bfl(flux2mi, ctx):
    """The same weights ingested under diffusers names (split q/k/v, [scale|shift] norm_out)
    give the same forward (WeightLoader.swift:397-480 vs :99-205)."""
    from flux2mi import synthetic as syn
    from flux2mi.api import DiT

    cfg = syn.DIT_PRESETS["tiny"]
    d = cfg["dim"]
    P = syn.make_dit_weights(cfg)
    D = {}
    ren = {"img_in.weight": "x_embedder.weight", "txt_in.weight": "context_embedder.weight",
           "time_in.in_layer.weight": "time_guidance_embed.timestep_embedder.linear_1.weight",
           "time_in.out_layer.weight": "time_guidance_embed.timestep_embedder.linear_2.weight",
           "double_stream_modulation_img.lin.weight": "double_stream_modulation_img.linear.weight",
           "double_stream_modulation_txt.lin.weight": "double_stream_modulation_txt.linear.weight",
           "single_stream_modulation.lin.weight": "single_stream_modulation.linear.weight",
           "final_layer.linear.weight": "proj_out.weight"}
    for k, v in P.items():
        if k in ren:
            D[ren[k]] = v
        elif k == "final_layer.adaLN_modulation.1.weight":
            D["norm_out.linear.weight"] = torch.cat([v[d:], v[:d]], dim=0)   # diffusers: [scale | shift]
        elif k.startswith("double_blocks."):
            i = k.split(".")[1]
            pre = f"transformer_blocks.{i}."
            tail = k.split(".", 2)[2]
            m = {"img_attn.proj.weight": "attn.to_out.0.weight", "txt_attn.proj.weight": "attn.to_add_out.weight",
                 "img_attn.norm.query_norm.scale": "attn.norm_q.weight", "img_attn.norm.key_norm.scale": "attn.norm_k.weight",
                 "txt_attn.norm.query_norm.scale": "attn.norm_added_q.weight", "txt_attn.norm.key_norm.scale": "attn.norm_added_k.weight",
                 "img_mlp.0.weight": "ff.linear_in.weight", "img_mlp.2.weight": "ff.linear_out.weight",
                 "txt_mlp.0.weight": "ff_context.linear_in.weight", "txt_mlp.2.weight": "ff_context.linear_out.weight"}
            if tail == "img_attn.qkv.weight":
                for n, part in zip(("to_q", "to_k", "to_v"), v.chunk(3, dim=0)):
                    D[pre + f"attn.{n}.weight"] = part.contiguous()
            elif tail == "txt_attn.qkv.weight":
                for n, part in zip(("add_q_proj", "add_k_proj", "add_v_proj"), v.chunk(3, dim=0)):
                    D[pre + f"attn.{n}.weight"] = part.contiguous()
            else:
                D[pre + m[tail]] = v
        elif k.startswith("single_blocks."):
            i = k.split(".")[1]
            pre = f"single_transformer_blocks.{i}.attn."
            tail = k.split(".", 2)[2]
            m = {"linear1.weight": "to_qkv_mlp_proj.weight", "linear2.weight": "to_out.weight",
                 "norm.query_norm.scale": "norm_q.weight", "norm.key_norm.scale": "norm_k.weight"}
            D[pre + m[tail]] = v
        else:
            raise AssertionError(k)
    inp = syn.make_inputs(cfg, 128, 128, s_txt=16)
    lat = R.pack_patchified_to_sequence(inp["noise"]).cuda()
    ii, ti = R.image_position_ids(128, 128).cuda(), R.text_position_ids(16).cuda()
    a = DiT(ctx, cfg).load(P).finalize()
    b = DiT(ctx, cfg).load(D).finalize()
    oa = a.forward(lat, inp["txt"].cuda(), 0.5, 0.0, ii, ti)
    ob = b.forward(lat, inp["txt"].cuda(), 0.5, 0.0, ii, ti)
    assert torch.equal(oa, ob)
    a.close(); b.close()


def test_missing_tensor_is_an_error(flux2mi, ctx):
    from flux2mi import synthetic as syn
    from flux2mi.api import DiT

    cfg = syn.DIT_PRESETS["tiny"]
    P = syn.make_dit_weights(cfg)
    P.pop("single_blocks.1.linear2.weight")
    d = DiT(ctx, cfg).load(P)
    with pytest.raises(flux2mi.Flux2Error) as e:
        d.finalize()
    assert "single_blocks.1.linear2.weight" in str(e.value)
    d.close()


def test_production_width_blocks(flux2mi, ctx):
    """Klein-4B width (d 3072, H 24, J 7680) with 1 double + 1 single block, 256x256 (S 768)."""
    cfg = dict(dim=3072, heads=24, n_double=1, n_single=1, joint_dim=7680, guidance=False)
    out, ref = _forward_both(flux2mi, ctx, cfg, s_txt=512, h=256, w=256)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 2e-2


def test_production_width_blocks_ragged(flux2mi, ctx):
    """The same width with nothing aligned to the 256-row GEMM tiles (s_txt 77, 272x256 -> 272 image tokens, S 349):
    the QK RMSNorm + RoPE runs in the qkv / linear1 GEMM epilogue here (>= 100 tiles), on partial row blocks, with
    non-trivial H / W / L positions feeding the in-kernel rotation angles."""
    cfg = dict(dim=3072, heads=24, n_double=1, n_single=1, joint_dim=7680, guidance=False)
    out, ref = _forward_both(flux2mi, ctx, cfg, s_txt=77, h=272, w=256)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 2e-2


# ---- weight_format 1: e4m3 block GEMMs ----------------------------------------------------------
def _fp8_case(ctx, cfg, s_txt, h, w, std=0.05, tweak=None):
    from flux2mi import synthetic as syn
    from flux2mi.api import DiT
    P = syn.make_dit_weights(cfg, seed=77, std=std)
    if tweak:
        tweak(P)
    inp = syn.make_inputs(cfg, h, w, s_txt=s_txt)
    lat = R.pack_patchified_to_sequence(inp["noise"])
    img_ids, txt_ids = R.image_position_ids(h, w), R.text_position_ids(s_txt)
    outs = {}
    for wf in ("bf16", "fp8"):
        dit = DiT(ctx, cfg, weights=wf).load(P).finalize()
        outs[wf] = dit.forward(lat.cuda(), inp["txt"].cuda(), 0.6, 4.0, img_ids.cuda(), txt_ids.cuda()).cpu()
        outs[wf + "_bytes"] = dit.param_bytes
        dit.close()
    args = (P, cfg, lat, inp["txt"].float(), torch.tensor([0.6]), None, img_ids, txt_ids)
    ref32 = R.dit_forward(*args)[0]
    with R.fp8_block_gemms():
        ref8 = R.dit_forward(*args)[0]
    return outs, ref32, ref8


@pytest.mark.parametrize("s_txt,h,w", [(16, 128, 128), (19, 80, 112), (64, 80, 112), (128, 128, 128)])
def test_tiny_forward_fp8(flux2mi, ctx, s_txt, h, w):
    """fp8 model vs the oracle running the same e4m3 emulation (weights at std 0.05 so that the
    block GEMMs matter: quantisation moves the output by q ~ 1.1e-2 rel-L2).  e4m3 rounding is
    discontinuous, so the bf16-level differences between the HIP path and the f32 oracle flip a
    few percent of the roundings and de-correlate part of the quantisation noise; the GEMM itself
    is checked to 1e-3 in test_gpu_ops.  Stated tolerances here:
      * rel-L2(HIP fp8, emulating oracle) <= 1e-2 and below q (the emulation explains the output
        better than the unquantised oracle does);
      * rel-L2(HIP fp8, f32 oracle) within [0.5 q, 1.5 q] (the noise level is the expected one);
      * (fp8 - bf16) of the HIP path points the same way as the oracle's (cosine > 0.7)."""
    from flux2mi import synthetic as syn
    # (text lengths 64 / 128: the block-scaled activations of proj / mlp_out / linear2 — 16 x 128 e4m3 blocks written by the SwiGLU and
    # attention epilogues — at tiny shapes: S = 64 + 35 = 99 ends in a partial row block inside a partial tile)
    outs, ref32, ref8 = _fp8_case(ctx, syn.DIT_PRESETS["tiny"], s_txt, h, w)
    assert torch.isfinite(outs["fp8"]).all()
    assert rel_l2(outs["bf16"], ref32) < 2e-2
    q = rel_l2(ref8, ref32)
    assert q > 5e-3                                      # the case is sensitive to quantisation at all
    e8 = rel_l2(outs["fp8"], ref8)
    assert e8 < 1e-2 and e8 < q
    assert 0.5 * q < rel_l2(outs["fp8"], ref32) < 1.5 * q
    d_hip = (outs["fp8"] - outs["bf16"]).flatten()
    d_ref = (ref8 - ref32).flatten()
    assert torch.dot(d_hip, d_ref) / (d_hip.norm() * d_ref.norm()) > 0.7
    assert outs["fp8_bytes"] < 0.7 * outs["bf16_bytes"]  # block weights are stored in one byte


def test_tiny_forward_fp8_large_key_norm_weights(flux2mi, ctx):
    """Key-norm weights x 64, query-norm weights / 64 (the same scores): the keys' bound sqrt(128) max|w_k| ~ 900 is beyond e4m3's 448, so
    the e4m3 keys of the attention are scaled DOWN (negative power of two, dit.cpp finalize / oracle fp8_qk) instead of saturating; the
    tolerances of test_tiny_forward_fp8 hold unchanged."""
    from flux2mi import synthetic as syn

    def tweak(P):
        for k in P:
            if k.endswith("key_norm.scale"):
                P[k] = (P[k].float() * 64.0).to(P[k].dtype)
            elif k.endswith("query_norm.scale"):
                P[k] = (P[k].float() / 64.0).to(P[k].dtype)
    outs, ref32, ref8 = _fp8_case(ctx, syn.DIT_PRESETS["tiny"], 16, 128, 128, tweak=tweak)
    assert torch.isfinite(outs["fp8"]).all()
    assert rel_l2(outs["bf16"], ref32) < 2e-2
    q = rel_l2(ref8, ref32)
    e8 = rel_l2(outs["fp8"], ref8)
    print(f"large key-norm weights: q {q:.3e}, HIP fp8 vs emulation {e8:.3e}")
    assert e8 < 1e-2 and e8 < q
    assert 0.5 * q < rel_l2(outs["fp8"], ref32) < 1.5 * q


def test_klein9b_width_blocks(flux2mi, ctx):
    """Klein-9B width (d 4096, 32 heads, J 12288), 1 double + 1 single block, S 3584: the N = 4096 GEMMs run on
    160-row tiles (5 MFMA row blocks per wave: 2 rounds of 5/8 tiles instead of 2 rounds at 56 %), the V^T GEMM
    among them with the permuted key order, next to the fused QK epilogue on 256-row tiles."""
    cfg = dict(dim=4096, heads=32, n_double=1, n_single=1, joint_dim=12288, guidance=False)
    out, ref = _forward_both(flux2mi, ctx, cfg, s_txt=512, h=768, w=1024)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 2e-2


def test_production_width_fused_paths_bf16_and_fp8(flux2mi, ctx):
    """Klein-4B width, 1 double + 1 single block, 512 text + 768x1024 image tokens (S 3584): large enough for every
    fused form of the production path — QK RMSNorm + RoPE in the GEMM epilogue and V^T written by its own
    role-swapped GEMM (12 x 14 = 168 tiles) — with bf16 and with e4m3 block GEMMs.  bf16 vs the f32 oracle
    <= 2e-2; fp8 vs the oracle running the same e4m3 emulation: closer to it than to the unquantised oracle."""
    cfg = dict(dim=3072, heads=24, n_double=1, n_single=1, joint_dim=7680, guidance=False)
    outs, ref32, ref8 = _fp8_case(ctx, cfg, 512, 768, 1024, std=0.02)
    assert torch.isfinite(outs["bf16"]).all() and torch.isfinite(outs["fp8"]).all()
    assert rel_l2(outs["bf16"], ref32) < 2e-2
    # e4m3 noise at this width (K 3072 ... 12288, weights at std 0.02): q = rel-L2(emulating oracle, f32 oracle) = 3.9e-2;
    # the HIP path sits 2.5e-2 from the emulation (roundings flipped by bf16-level differences de-correlate part of the
    # noise, see test_tiny_forward_fp8) — the same value with the fused epilogue / V^T GEMM switched off
    q = rel_l2(ref8, ref32)
    e8 = rel_l2(outs["fp8"], ref8)
    print(f"production width fp8: q {q:.3e}, HIP vs emulation {e8:.3e}, HIP vs f32 oracle {rel_l2(outs['fp8'], ref32):.3e}")
    assert e8 < 3e-2 and e8 < q
    assert 0.5 * q < rel_l2(outs["fp8"], ref32) < 1.5 * q


def test_fp8_rejects_unsupported_dim(flux2mi, ctx):
    """dim must be a multiple of 256 for the fp8 tiles (include/flux2mi.h weight_format)."""
    from flux2mi import Flux2Error
    from flux2mi.api import DiT
    cfg = dict(dim=128, heads=1, n_double=1, n_single=1, joint_dim=128, guidance=False)
    with pytest.raises(Flux2Error):
        DiT(ctx, cfg, weights="fp8")


# ---- KV-cached reference conditioning (klein-9b-kv path) ------------------------
def _kv_case(ctx, cfg, s_txt, h, w, ref_hw, weights="bf16", std=0.05):
    from flux2mi import synthetic as syn
    from flux2mi.api import DiT, KVCache
    P = syn.make_dit_weights(cfg, seed=5, std=std)
    inp = syn.make_inputs(cfg, h, w, s_txt=s_txt)
    lat = R.pack_patchified_to_sequence(inp["noise"])
    g = torch.Generator().manual_seed(1)
    s_ref = ref_hw[0] * ref_hw[1]
    ref = torch.randn((1, s_ref, 128), generator=g)
    lat2 = lat + 0.3 * torch.randn(lat.shape, generator=g)
    ref_ids = R.reference_position_ids([ref_hw[0]], [ref_hw[1]])
    img_ids, txt_ids = R.image_position_ids(h, w), R.text_position_ids(s_txt)
    gt = torch.tensor([4.0]) if cfg["guidance"] else None
    # oracle: extraction at t0, cached pass at t1 on moved latents
    o0, cache = R.dit_forward_kv(P, cfg, lat, inp["txt"].float(), torch.tensor([0.8]), gt, img_ids, txt_ids, ref_lat=ref, ref_ids=ref_ids)
    o1, _ = R.dit_forward_kv(P, cfg, lat2, inp["txt"].float(), torch.tensor([0.55]), gt, img_ids, txt_ids, cache=cache)
    dit = DiT(ctx, cfg, weights=weights).load(P).finalize()
    kv = KVCache(dit, s_txt, s_ref)
    d = lambda x: x.cuda()
    h0 = dit.forward_kv_extract(kv, d(lat), d(ref), d(inp["txt"]), 0.8, 4.0, d(img_ids), d(txt_ids), d(ref_ids)).cpu()
    h1 = dit.forward_kv_cached(kv, d(lat2), d(inp["txt"]), 0.55, 4.0, d(img_ids), d(txt_ids)).cpu()
    h0_again = dit.forward_kv_cached(kv, d(lat), d(inp["txt"]), 0.8, 4.0, d(img_ids), d(txt_ids)).cpu()
    kv.close()
    dit.close()
    return (o0[0], o1[0]), (h0, h1, h0_again)


@pytest.mark.parametrize("preset,s_txt,h,w,ref_hw", [
    ("tiny", 16, 64, 64, (4, 6)),      # s_ref 24: the V^T key groups straddle the ref/out boundary
    ("tiny", 19, 80, 112, (6, 10)),    # nothing aligned: s_txt 19, s_ref 60, s_out 35
    ("tiny-g", 16, 64, 64, (4, 4)),    # guidance embedding
    ("tiny", 32, 128, 128, (8, 8)),    # aligned, more than one attention q-block per range
])
def test_kv_extract_and_cached_vs_oracle(flux2mi, ctx, preset, s_txt, h, w, ref_hw):
    """forwardKVExtract / forwardKVCached vs the oracle's masked / concatenated-KV restatement
    (rel-L2 <= 2e-2), and the structural identity that a cached pass on the extraction pass's own
    inputs reproduces its output (same keys, same values: <= 2e-3, bf16 rounding of split launches)."""
    from flux2mi import synthetic as syn
    (o0, o1), (h0, h1, h0_again) = _kv_case(ctx, syn.DIT_PRESETS[preset], s_txt, h, w, ref_hw)
    assert torch.isfinite(h0).all() and torch.isfinite(h1).all()
    assert rel_l2(h0, o0) < 2e-2
    assert rel_l2(h1, o1) < 2e-2
    assert rel_l2(h0_again, h0) < 2e-3


def test_kv_extract_and_cached_production_width(flux2mi, ctx):
    """KV extraction / cached passes at Klein-4B width: the fused QK norm + RoPE epilogue on the two live row ranges
    of a cached pass ([txt] and [out] with the reference gap between them: table rows offset per range)."""
    cfg = dict(dim=3072, heads=24, n_double=1, n_single=1, joint_dim=7680, guidance=False)
    (o0, o1), (h0, h1, h0_again) = _kv_case(ctx, cfg, 64, 256, 256, (16, 16))
    assert torch.isfinite(h0).all() and torch.isfinite(h1).all()
    assert rel_l2(h0, o0) < 2e-2
    assert rel_l2(h1, o1) < 2e-2
    assert rel_l2(h0_again, h0) < 2e-3


@pytest.mark.parametrize("s_txt,h,w,ref_hw", [(64, 256, 256, (9, 15)), (19, 128, 192, (7, 9))])
def test_kv_extract_unaligned_reference_rows_production_width(flux2mi, ctx, s_txt, h, w, ref_hw):
    """Klein-4B width, two double + two single blocks, s_txt + s_ref NOT a multiple of the 64-key tile (64 + 135 = 199, 19 + 63 = 82):
    the reference queries of the extraction pass run against a key prefix that ends inside a KV tile, with the output tokens' V^T
    columns right behind it.  Checked where a leak would land: the cached K / V of every block after the first are functions of the
    reference rows' attention output of the blocks before — compared PER REFERENCE ROW with the oracle's cache
    (`LayerKVCacheEntry.keys / .values`, TransformerKVCache.swift:13-25), then the cached pass against the oracle."""
    from flux2mi import synthetic as syn
    from flux2mi.api import DiT, KVCache
    cfg = dict(dim=3072, heads=24, n_double=2, n_single=2, joint_dim=7680, guidance=False)
    P = syn.make_dit_weights(cfg, seed=5, std=0.05)
    inp = syn.make_inputs(cfg, h, w, s_txt=s_txt)
    lat = R.pack_patchified_to_sequence(inp["noise"])
    g = torch.Generator().manual_seed(3)
    s_ref = ref_hw[0] * ref_hw[1]
    assert (s_txt + s_ref) % 64 != 0
    ref = torch.randn((1, s_ref, 128), generator=g)
    lat2 = lat + 0.3 * torch.randn(lat.shape, generator=g)
    ref_ids = R.reference_position_ids([ref_hw[0]], [ref_hw[1]])
    img_ids, txt_ids = R.image_position_ids(h, w), R.text_position_ids(s_txt)
    o0, cache = R.dit_forward_kv(P, cfg, lat, inp["txt"].float(), torch.tensor([0.8]), None, img_ids, txt_ids, ref_lat=ref, ref_ids=ref_ids)
    o1, _ = R.dit_forward_kv(P, cfg, lat2, inp["txt"].float(), torch.tensor([0.55]), None, img_ids, txt_ids, cache=cache)
    dit = DiT(ctx, cfg).load(P).finalize()
    kv = KVCache(dit, s_txt, s_ref)
    d = lambda x: x.cuda()
    h0 = dit.forward_kv_extract(kv, d(lat), d(ref), d(inp["txt"]), 0.8, 4.0, d(img_ids), d(txt_ids), d(ref_ids)).cpu()
    worst = 0.0
    for layer in range(4):
        keys, vals = kv.entry(layer)
        ok, ov = cache[layer]                                   # [1, H, s_ref, 128]
        ok = ok[0].permute(1, 0, 2).reshape(s_ref, -1)
        ov = ov[0].permute(1, 0, 2).reshape(s_ref, -1)
        for got, exp in ((keys.float().cpu(), ok), (vals.float().cpu(), ov)):
            row_err = (got - exp).norm(dim=1) / exp.norm(dim=1)
            worst = max(worst, float(row_err.max()))
            assert float(row_err.max()) < 3e-2, (layer, float(row_err.max()), int(row_err.argmax()))
    h1 = dit.forward_kv_cached(kv, d(lat2), d(inp["txt"]), 0.55, 4.0, d(img_ids), d(txt_ids)).cpu()
    kv.close()
    dit.close()
    assert rel_l2(h0, o0[0]) < 2e-2 and rel_l2(h1, o1[0]) < 2e-2


def test_kv_cached_before_extract_is_an_error(flux2mi, ctx):
    """The reference calls fatalError on a missing cache entry (Flux2Transformer.swift:488); here: a status."""
    from flux2mi import Flux2Error, synthetic as syn
    from flux2mi.api import DiT, KVCache
    cfg = syn.DIT_PRESETS["tiny"]
    dit = DiT(ctx, cfg).load(syn.make_dit_weights(cfg)).finalize()
    kv = KVCache(dit, 16, 16)
    lat = torch.zeros((16, 128), device="cuda")
    txt = torch.zeros((16, cfg["joint_dim"]), dtype=torch.bfloat16, device="cuda")
    ids = torch.zeros((16, 4), dtype=torch.int32, device="cuda")
    with pytest.raises(Flux2Error):
        dit.forward_kv_cached(kv, lat, txt, 0.5, 4.0, ids, ids)
    kv.close()
    dit.close()


def test_kv_fp8_runs_and_matches_emulation(flux2mi, ctx):
    """The KV paths share the fp8 block GEMMs: cached pass vs the emulating oracle."""
    from flux2mi import synthetic as syn
    cfg = syn.DIT_PRESETS["tiny"]
    with R.fp8_block_gemms():
        (o0, o1), (h0, h1, _) = _kv_case(ctx, cfg, 16, 64, 64, (4, 4), weights="fp8")
    assert rel_l2(h0, o0) < 2e-2 and rel_l2(h1, o1) < 2e-2


# ---- attention softmax offset: fixed zero offset vs per-query data offset ----------------------------
@pytest.mark.parametrize("factor,s_txt,h,w,fixed", [(1.0, 16, 128, 128, True), (1.5, 19, 80, 112, True), (2.0, 16, 128, 128, False), (4.0, 19, 80, 112, False)])
def test_forward_with_scaled_qk_norm_weights(flux2mi, ctx, factor, s_txt, h, w, fixed):
    """The DiT's attention takes a fixed zero softmax offset when the QK-RMSNorm weights bound the pre-scaled scores tightly
    enough (16.8 max|w_q| max|w_k| <= 80 in the log2 domain), and a per-query offset from the data otherwise (attention64b.hip,
    DOFF).  The synthetic norm scales are 1 +- 0.1 (max ~1.3): factor 1.0 / 1.5: fixed offset in every block (bounds ~29 / ~65);
    2.0 / 4.0: data offset in every block (~110 / ~460, much peakier softmaxes).  All against the oracle, rel-L2 <= 2e-2; the library reports which form each block takes."""
    from flux2mi import synthetic as syn
    from flux2mi.api import DiT
    cfg = syn.DIT_PRESETS["tiny"]
    P = syn.make_dit_weights(cfg, seed=1234, std=0.05)
    for k in list(P):
        if k.endswith("norm.query_norm.scale") or k.endswith("norm.key_norm.scale"):
            P[k] = (P[k].float() * factor).to(torch.bfloat16)
    inp = syn.make_inputs(cfg, h, w, s_txt=s_txt)
    lat = R.pack_patchified_to_sequence(inp["noise"])
    img_ids, txt_ids = R.image_position_ids(h, w), R.text_position_ids(s_txt)
    dit = DiT(ctx, cfg).load(P).finalize()
    paths = dit.attention_softmax_paths()
    out = dit.forward(lat.cuda(), inp["txt"].cuda(), 0.731, 4.0, img_ids.cuda(), txt_ids.cuda()).cpu()
    dit.close()
    ref = R.dit_forward(P, cfg, lat, inp["txt"].float(), torch.tensor([0.731]), None, img_ids, txt_ids)[0]
    n_blocks = cfg["n_double"] + cfg["n_single"]
    assert paths == ({"fixed_offset_blocks": n_blocks, "online_blocks": 0} if fixed else {"fixed_offset_blocks": 0, "online_blocks": n_blocks}), paths
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 2e-2


def test_online_mode_forward_and_kv_paths(flux2mi, ctx):
    """`set_attention_softmax("online")` (bench.py --attn-softmax online): no score bound assumed in any block.  The plain forward at
    Klein-4B width and the KV extraction / cached passes (query ranges that see a subset of the keys, cached K / V^T) against the
    oracle, and within bf16 rounding of the default mode on the same weights."""
    from flux2mi import synthetic as syn
    from flux2mi.api import DiT, KVCache
    cfg = dict(dim=3072, heads=24, n_double=1, n_single=1, joint_dim=7680, guidance=False)
    s_txt, h, w, ref_hw = 64, 256, 256, (16, 16)
    P = syn.make_dit_weights(cfg, seed=5, std=0.05)
    inp = syn.make_inputs(cfg, h, w, s_txt=s_txt)
    lat = R.pack_patchified_to_sequence(inp["noise"])
    g = torch.Generator().manual_seed(1)
    s_ref = ref_hw[0] * ref_hw[1]
    ref = torch.randn((1, s_ref, 128), generator=g)
    lat2 = lat + 0.3 * torch.randn(lat.shape, generator=g)
    ref_ids = R.reference_position_ids([ref_hw[0]], [ref_hw[1]])
    img_ids, txt_ids = R.image_position_ids(h, w), R.text_position_ids(s_txt)
    o_plain = R.dit_forward(P, cfg, lat, inp["txt"].float(), torch.tensor([0.731]), None, img_ids, txt_ids)[0]
    o0, cache = R.dit_forward_kv(P, cfg, lat, inp["txt"].float(), torch.tensor([0.8]), None, img_ids, txt_ids, ref_lat=ref, ref_ids=ref_ids)
    o1, _ = R.dit_forward_kv(P, cfg, lat2, inp["txt"].float(), torch.tensor([0.55]), None, img_ids, txt_ids, cache=cache)
    d = lambda x: x.cuda()
    outs = {}
    for mode in ("auto", "online"):
        dit = DiT(ctx, cfg).load(P).finalize()
        dit.set_attention_softmax(mode)
        n_blocks = cfg["n_double"] + cfg["n_single"]
        assert dit.attention_softmax_paths() == ({"fixed_offset_blocks": n_blocks, "online_blocks": 0} if mode == "auto"
                                                 else {"fixed_offset_blocks": 0, "online_blocks": n_blocks})
        kv = KVCache(dit, s_txt, s_ref)
        hp = dit.forward(d(lat), d(inp["txt"]), 0.731, 4.0, d(img_ids), d(txt_ids)).cpu()
        h0 = dit.forward_kv_extract(kv, d(lat), d(ref), d(inp["txt"]), 0.8, 4.0, d(img_ids), d(txt_ids), d(ref_ids)).cpu()
        h1 = dit.forward_kv_cached(kv, d(lat2), d(inp["txt"]), 0.55, 4.0, d(img_ids), d(txt_ids)).cpu()
        kv.close()
        dit.close()
        outs[mode] = (hp, h0, h1)
    for got, want in zip(outs["online"], (o_plain, o0[0], o1[0])):
        assert torch.isfinite(got).all()
        assert rel_l2(got, want) < 2e-2
    for a_, b_ in zip(outs["online"], outs["auto"]):
        assert rel_l2(a_, b_) < 5e-3


# ---- edge sizes ------------------------------------------------------------------------------------
def test_maximum_size_2048_forward(flux2mi, ctx):
    """The reference's largest output, 2048 x 2048 (S_img 16384, S 16400): long KV loops (257 tiles),
    the 8-wave attention workgroups with the fixed-offset softmax, GEMMs with M = 16400.  Tiny widths
    so that the f32 oracle stays affordable on the CPU; rel-L2 <= 2e-2."""
    from flux2mi import synthetic as syn
    out, ref = _forward_both(flux2mi, ctx, syn.DIT_PRESETS["tiny"], s_txt=16, h=2048, w=2048)
    assert out.shape == (16384, 128)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 2e-2


def test_three_reference_images(flux2mi, ctx):
    """I2I with the maximum of three reference images (Flux2Pipeline.swift:921), T offsets 10, 20, 30
    (LatentUtils.swift:324-346), different sizes, appended after the output tokens."""
    from flux2mi import synthetic as syn
    g = torch.Generator().manual_seed(6)
    hs, ws = [4, 2, 6], [6, 8, 4]
    n = sum(a * b for a, b in zip(hs, ws))
    ref_lat = torch.randn((1, n, 128), generator=g)
    ref_ids = R.reference_position_ids(hs, ws)
    assert ref_ids.shape == (n, 4) and set(ref_ids[:, 0].tolist()) == {10, 20, 30}
    out, ref = _forward_both(flux2mi, ctx, syn.DIT_PRESETS["tiny"], s_txt=16, h=96, w=64, refs=(ref_lat, ref_ids))
    s_out = (96 // 16) * (64 // 16)
    assert rel_l2(out[:s_out], ref[:s_out]) < 2e-2


def test_smallest_image(flux2mi, ctx):
    """16 x 16 pixels = ONE image token (+ 1 text token): S = 2."""
    from flux2mi import synthetic as syn
    out, ref = _forward_both(flux2mi, ctx, syn.DIT_PRESETS["tiny"], s_txt=1, h=16, w=16)
    assert out.shape == (1, 128)
    assert rel_l2(out, ref) < 2e-2
