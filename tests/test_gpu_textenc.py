"""-m gpu: Qwen3 text encoder -> Klein prompt embeddings, HIP path vs the oracle
(whose Qwen3 restatement is pinned against HF transformers: tests/test_textenc_oracle.py) and vs the
committed HF fixture directly.

Tolerance (stated): rel-L2 <= 2e-2 per extracted hidden state (bf16 GEMM operands, bf16 q/k after the
per-head RMSNorm, f32 residual stream), same as the DiT forward."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import flux2_ref as R
from tests.util import rel_l2

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden_qwen3 as G  # noqa: E402

pytestmark = pytest.mark.gpu


def _encoder(ctx, tcfg, P):
    from flux2mi.textenc import TextEncoder
    return TextEncoder(ctx, tcfg).load(P).finalize()


def test_tiny_vs_hf_fixture(flux2mi, ctx):
    g = np.load("tests/golden/qwen3_tiny.npz")
    P = G.make_qwen3_weights(G.TINY)
    te = _encoder(ctx, G.TINY, P)
    out = te.forward(torch.from_numpy(g["ids"]).to(torch.int32).cuda(), int(g["valid_len"][0])).float().cpu()[0]
    assert out.shape == (48, 768)
    for i in range(3):
        assert rel_l2(out[:, 256 * i:256 * (i + 1)], torch.from_numpy(g[f"h{i + 1}"])) < 2e-2
    te.close()


@pytest.mark.parametrize("S,valid", [(512, 37), (512, 512), (200, 1), (130, 129)])
def test_lengths_and_padding_vs_oracle(flux2mi, ctx, S, valid):
    """Production sequence length 512 with a short prompt, a full one, a single token, and a ragged
    length; pad positions included (the DiT attends to them)."""
    P = G.make_qwen3_weights(G.TINY)
    te = _encoder(ctx, G.TINY, P)
    gen = torch.Generator().manual_seed(S + valid)
    ids = torch.randint(0, G.TINY["vocab"], (S,), generator=gen)
    ids[valid:] = 3
    out = te.forward(ids.to(torch.int32).cuda(), valid).float().cpu()
    ref = R.klein_embeddings(P, G.TINY, ids, valid)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 2e-2
    assert rel_l2(out[0, :valid], ref[0, :valid]) < 2e-2
    te.close()


def test_production_width_two_layers(flux2mi, ctx):
    """Qwen3-4B widths (hidden 2560, 32 / 8 heads of 128, MLP 9728) with 3 layers, extraction after
    1, 2, 3: the GEMM shapes and the GQA attention of the real encoder."""
    tcfg = dict(vocab=2048, hidden=2560, intermediate=9728, n_layers=4, heads=32, kv_heads=8, head_dim=128,
                rope_theta=1.0e6, rms_eps=1.0e-6, extract_layers=(1, 2, 3))
    P = G.make_qwen3_weights(tcfg, seed=9)
    te = _encoder(ctx, tcfg, P)
    gen = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 2048, (512,), generator=gen)
    ids[60:] = 0
    out = te.forward(ids.to(torch.int32).cuda(), 60).float().cpu()
    ref = R.klein_embeddings(P, tcfg, ids, 60)
    assert out.shape == (1, 512, 7680)
    assert rel_l2(out, ref) < 2e-2
    te.close()


def test_prompt_to_image_tiny(flux2mi, ctx):
    """ids -> text encoder -> DiT -> VAE: the encoder's output is accepted as `precomputed_embeddings`
    (3 * hidden == joint_dim), end to end on the GPU, and matches the oracle chain."""
    from flux2mi import synthetic as syn
    from flux2mi.pipeline import Flux2Pipeline
    tcfg = dict(G.TINY, hidden=128, extract_layers=(1, 2, 3))      # 3 * 128 = 384 = tiny joint_dim
    P = G.make_qwen3_weights(tcfg, seed=4)
    te = _encoder(ctx, tcfg, P)
    ids, n = flux2mi.textenc.pad_klein_token_ids([11, 12, 13, 14, 15, 16, 17], 16, pad_id=3)
    emb = te.extract_klein_embeddings("", max_length=16, token_ids=[11, 12, 13, 14, 15, 16, 17])
    # pad id differs from the helper default here (vocab 1000): redo with the explicit ids
    emb = te.forward(torch.tensor(ids, dtype=torch.int32).cuda(), n)
    cfg, vcfg = syn.DIT_PRESETS["tiny"], syn.VAE_PRESETS["tiny"]
    DP = syn.make_dit_weights(cfg)
    pipe = Flux2Pipeline(model=cfg, vae_variant=vcfg).load_models(DP, syn.make_vae_weights(vcfg))
    inp = syn.make_inputs(cfg, 64, 64, s_txt=16)
    res = pipe.generate_with_result("textToImage", height=64, width=64, steps=2, precomputed_embeddings=emb,
                                    noise=inp["noise"], return_latents=True)
    ref_emb = R.klein_embeddings(P, tcfg, torch.tensor(ids), n)
    gen = R.generate(DP, cfg, None, None, ref_emb, inp["noise"], 64, 64, 2, decode=False)
    assert rel_l2(res.latents.cpu(), gen["final_latents"]) < 3e-2
    te.close()


def test_errors(flux2mi, ctx):
    from flux2mi import Flux2Error
    from flux2mi.textenc import TextEncoder
    with pytest.raises(Flux2Error):   # head_dim 80 (the reference's fallback default) is not built
        TextEncoder(ctx, dict(G.TINY, head_dim=80))
    te = TextEncoder(ctx, G.TINY)
    with pytest.raises(Flux2Error):   # not finalized
        te.forward(torch.zeros(8, dtype=torch.int32, device="cuda"), 4)
    te.close()


# ---- Dev's encoder: Mistral (no q/k norm), LEFT padding ----------------------------------------------
def test_mistral_tiny_vs_hf_fixture_and_oracle(flux2mi, ctx):
    g = np.load("tests/golden/mistral_tiny.npz")
    P = G.make_qwen3_weights(G.MISTRAL_TINY, seed=78)
    te = _encoder(ctx, G.MISTRAL_TINY, P)
    v, pad = int(g["valid_len"][0]), int(g["pad_left"][0])
    ids = torch.from_numpy(g["ids"])
    out = te.forward(ids.to(torch.int32).cuda(), v, pad_left=pad).float().cpu()[0]
    ref = R.klein_embeddings(P, G.MISTRAL_TINY, ids, v, pad_left=pad)[0]
    assert torch.isfinite(out).all()
    for i in range(3):   # real rows against HF, every row (pads included) against the oracle
        assert rel_l2(out[pad:, 256 * i:256 * (i + 1)], torch.from_numpy(g[f"h{i + 1}"][pad:])) < 2e-2
    assert rel_l2(out, ref) < 2e-2
    assert rel_l2(out[:pad], ref[:pad]) < 2e-2
    te.close()


@pytest.mark.parametrize("S,valid", [(512, 60), (512, 449), (512, 512), (200, 3), (130, 1)])
def test_left_padding_lengths_vs_oracle(flux2mi, ctx, S, valid):
    """Left padding across the attention tiling: pad/real boundary inside a 64-key tile, inside a
    128-row workgroup, workgroups entirely in the pads, no padding at all."""
    P = G.make_qwen3_weights(G.MISTRAL_TINY, seed=78)
    te = _encoder(ctx, G.MISTRAL_TINY, P)
    gen = torch.Generator().manual_seed(S + valid)
    ids = torch.randint(0, G.MISTRAL_TINY["vocab"], (S,), generator=gen)
    pad = S - valid
    ids[:pad] = 11
    out = te.forward(ids.to(torch.int32).cuda(), valid, pad_left=pad).float().cpu()
    ref = R.klein_embeddings(P, G.MISTRAL_TINY, ids, valid, pad_left=pad)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 2e-2
    assert rel_l2(out[0, pad:], ref[0, pad:]) < 2e-2
    if pad:
        assert rel_l2(out[0, :pad], ref[0, :pad]) < 2e-2
    te.close()


def test_mistral_width_slice(flux2mi, ctx):
    """Mistral-Small widths for attention (hidden 5120, 32 / 8 heads of 128) with a narrowed MLP, 2 of 3
    layers run, through `extract_flux_embeddings` (left padding to 512)."""
    tcfg = dict(vocab=2048, hidden=5120, intermediate=4096, n_layers=4, heads=32, kv_heads=8, head_dim=128,
                rope_theta=1.0e9, rms_eps=1.0e-5, extract_layers=(1, 2, 3), qk_norm=0)
    P = G.make_qwen3_weights(tcfg, seed=10)
    te = _encoder(ctx, tcfg, P)
    toks = list(range(100, 171))
    out = te.extract_flux_embeddings(toks, 512, pad_id=11).float().cpu()
    ids = torch.tensor([11] * (512 - len(toks)) + toks)
    ref = R.klein_embeddings(P, tcfg, ids, len(toks), pad_left=512 - len(toks))
    assert out.shape == (1, 512, 15360)
    assert rel_l2(out, ref) < 2e-2
    te.close()
